"""Record the fastest rocBLAS / hipBLASLt solution for every GEMM shape of the 4096-env rollout step with torch's TunableOp and
write kinpoly_amd/assets/tunableop_gfx950.csv (read by kinpoly_amd.nets.enable_tuned_gemms).  Run on the GPU box:
    python tools/tune_gemms.py && cp gpurun_out/tune/tunableop_gfx950_0.csv kinpoly_amd/assets/tunableop_gfx950.csv
The update's shapes (98 304-row MLP GEMMs, the recurrent [4096, 1024] x [1024, 3072] addmm with bias, the weight-gradient GEMMs) are recorded by
    KP_TUNE=1 KP_ITERS=1 python tools/update_profile.py        (writes gpurun_out/tune/tunableop_update_0.csv; merge the new rows into the asset)
Round 3: the recurrent forward GEMM alone went 267 -> 183 us (it ran on the library's default pick), T_update 0.957 -> 0.846 s at 4096 x 24."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "gpurun_out", "tune")
os.makedirs(out, exist_ok=True)
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="100",
                  PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="10", PYTORCH_TUNABLEOP_FILENAME=os.path.join(out, "tunableop_gfx950_%d.csv"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

import kinpoly_amd.nets as nets  # noqa: E402

nets.enable_tuned_gemms = lambda *a, **k: False          # tune from scratch
env, policy, sampler, std = bench.build_engine(0, 4, 64)
bench.rollout_steps(sampler, 4)
torch.cuda.synchronize()
print("TunableOp writes", os.path.join(out, "tunableop_gfx950_0.csv"), "when this process exits; copy it to kinpoly_amd/assets/tunableop_gfx950.csv")
