# round 5, sixth GPU call: kernel tables of the headline under the selection rules 1 and 2, then the A/B again
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 1 2; do
  export EMCID_SP16_MFMA16=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_prof_m$v -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-stage0 --no-variants --no-gemm-ab > gpurun_out/r05_prof_m$v.json 2> gpurun_out/r05_prof_m$v.err; echo "prof $v rc $?"
  f=$(find gpurun_out/r05_prof_m$v -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_m${v}_kernel_stats.csv; rm -rf gpurun_out/r05_prof_m$v
done
unset EMCID_SP16_MFMA16
out=gpurun_out/mfma16_ab3; mkdir -p $out
for i in 1 2; do for v in 1 2; do
  EMCID_SP16_MFMA16=$v timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-stage0 --no-cpu-baseline --no-gemm-ab > $out/v${v}_$i.json 2> $out/v${v}_$i.err
done; done
python - > gpurun_out/r05_mfma16_ab3.txt <<PY
import json
for i in (1,2):
    for v in (1,2):
        d=json.loads(open(f"gpurun_out/mfma16_ab3/v{v}_{i}.json").read().strip().splitlines()[-1])
        pc=sorted(d["ms_per_call"])
        print("MFMA16",v,"run",i,"mean",round(d["ms_per_step"],3),"median",round(d["ms_per_call_median"],3),"device",round(d["device_ms_per_step"],3),"linear",round(d["kernel_classes"]["linear"]["ms_per_step"],3),"frac",round(d["roofline"]["frac"],3),
              "n100",round(d["n100"]["ms_per_call_median"],3),"realistic",round(d["realistic_names"]["ms_per_call_median"],2),"n1500",round(d["n1500"]["ms_per_call_median"],2),"sdxl",round(d["sdxl"]["ms_per_call_median"],2),"nsp",round(d["no_shared_prefix"]["ms_per_call_median"],2))
PY
cat gpurun_out/r05_mfma16_ab3.txt
echo done
