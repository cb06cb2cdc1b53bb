# round 6: the N x N chain with and without the shadow product P = Yt X in its leaf launches, same box, alternating processes
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for m in 1 0; do
  EMCID_SHADOW_P=$m timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-stage0 --no-cpu-baseline --no-gemm-ab --no-variants > gpurun_out/r06_shadow${m}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_shadow${m}_$rep.json").read().strip().splitlines()[-1])
k=d["kernel_classes"]
print("EMCID_SHADOW_P=$m rep $rep: ms", round(d["ms_per_call_median"],3), "device", round(d["device_ms_per_step"],3), "solve", round(d["solve"]["ms_per_step"],3),
      {c: round(k[c]["ms_per_step"],3) for c in ("chol_leaf","chol_panel","inv_apply","delta_w","trsm_diag","assemble") if c in k}, "leaf launches", k["chol_leaf"]["launches_per_step"])
PY
done; done
