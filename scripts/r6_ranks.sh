# round 6: bench.py's multi-rank path on the one-GPU box (ranks share cuda:0, gloo collectives staged through the host): the LINE's
# shape for N > 1 (strong-scaling value, weak_replicas, per_rank phases) — functional evidence, not a scaling measurement
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for n in 2 4; do
  EMCID_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus $n --steps 8 --warmup 3 > gpurun_out/r06_ranks${n}_gloo_shared_gpu.json 2> gpurun_out/r06_ranks${n}.err; echo "ranks $n rc $?"
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_ranks${n}_gloo_shared_gpu.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("n_gpus","scaling","value","ms_per_step")}, "weak", (d.get("weak_replicas") or {}).get("value"))
pr=d.get("per_rank") or []
for r in pr if isinstance(pr, list) else []:
    print("  rank", r["rank"], "concepts", r["concepts_of_this_rank"], "ms/call", round(r["ms_per_call"],2), "collectives", round(r["collectives_ms_per_call"],2))
PY
done
