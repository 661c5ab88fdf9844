set -e
cd $GRAFT_REPO_ROOT
T=$(mktemp -d)
python - <<PY
import sys, numpy as np
sys.path.insert(0, "point-cloud-compression_amd")
from pccx import plyio, synth
import os
os.makedirs("$T/data/train", exist_ok=True)
for i in range(3):
    plyio.save_point_cloud(synth.cad_cloud(40 + i, 8192) * np.float32(2.0), "$T/data/train/c%d.ply" % i)
PY
for extra in "--eager" "--autocast" "--eager --autocast" ""; do
  python point-cloud-compression_amd/cli/train.py --train_glob "$T/data/**/train/*.ply" --model_save_folder $T/m --max_steps 5 --step_window 3 --reset --rate_loss_enable_step 2 --lamda 10 $extra 2>&1 | grep -E "Step|Resetting|Error|error" | head -4
  echo "== done: $extra"
done
