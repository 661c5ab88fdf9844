import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/point-cloud-compression_amd")
import tests.test_train_step as T
import pccx
which = sys.argv[1].split(",")
for name in which:
    f = getattr(T, name)
    try:
        if name == "test_training_step_matches_autograd_and_adam":
            f("chamfer"); f("hybrid")
        else:
            f()
        print("ran", name, flush=True)
    except AssertionError as e:
        print("assert in", name, str(e)[:200], flush=True)
sys.argv = [sys.argv[0], "4"]
exec(open("/root/repo/tools/experiments/r3/dbg_train.py").read())
