"""CPU oracle for the LICOS learned-compression hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``licos_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

PARITY UNPINNED: the arithmetic restated here lives in CompressAI, a
third-party, un-pinned pip dependency of the reference
(/root/reference/environment.yml:27-29) that is neither vendored in the
reference tree nor installed in this image.  The reference's own tests hold
no golden vectors for this path.  The restatement
follows CompressAI's published algorithm (files named per function) and the
reference's call sites; the convolutions call torch's CPU conv2d /
conv_transpose2d, which *is* the reference's CPU arithmetic.
"""
