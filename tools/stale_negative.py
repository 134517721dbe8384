"""Dev check: with the weights epoch disabled (the round-1 behaviour) the stale-cache tests must FAIL."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from licos_amd import ops
ops.touch_weights = lambda: None
import test_gpu_stale as t
for fn in (t.test_eval_after_fused_adam_steps_sees_the_new_weights, t.test_scaled_flat_state_is_seen_by_the_next_forward):
    try:
        fn()
        print("NOT RED:", fn.__name__)
    except AssertionError as e:
        print("red as expected:", fn.__name__, str(e)[:80].replace("\n", " "))
