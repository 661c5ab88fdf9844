import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "point-cloud-compression_amd"))
import tests.test_train_step as t
for i in range(8):
    try:
        t.test_autocast_run_of_twenty_steps_stays_in_a_band_of_the_fp32_run()
        print("run", i, "ok", flush=True)
    except AssertionError as e:
        print("run", i, "FAIL", str(e)[:200], flush=True)
