# round 5, fifth GPU call: the 64 x 64 form and the new selection rule: parity, small-shape microbenchmark, headline / n100 A/B
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear_sp16_vs_torch or producers or trie_forward" > gpurun_out/r05_t5.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r05_t5.txt
tail -4 gpurun_out/r05_t5.txt
MB_DBG=0 MB_FORMS=256,448,512 MB_SHAPES=qkv-n100,fc1-n100,out-n100,fc2-n100,fc2-keys,out-query,fc1-query,fc2-query,fc2-keys-bigG,qkv-2k,fc2-2k timeout -k 10 300 python scripts/mb_linear_sp16_r5.py > gpurun_out/r05_mb_linear_sp16_small.txt 2>&1; echo "mb rc $?"
out=gpurun_out/mfma16_ab2; mkdir -p $out
for i in 1 2; do for v in 1 2; do
  EMCID_SP16_MFMA16=$v timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-stage0 --no-cpu-baseline --no-gemm-ab > $out/v${v}_$i.json 2> $out/v${v}_$i.err
done; done
python - > gpurun_out/r05_mfma16_ab2.txt <<PY
import json
for i in (1,2):
    for v in (1,2):
        d=json.loads(open(f"gpurun_out/mfma16_ab2/v{v}_{i}.json").read().strip().splitlines()[-1])
        pc=sorted(d["ms_per_call"])
        print("MFMA16",v,"run",i,"mean",round(d["ms_per_step"],3),"median",round(d["ms_per_call_median"],3),"p10",round(pc[6],2),"p90",round(pc[53],2),"device",round(d["device_ms_per_step"],3),"linear",round(d["kernel_classes"]["linear"]["ms_per_step"],3),"frac",round(d["roofline"]["frac"],3),
              "n100",round(d["n100"]["ms_per_call_median"],3),"realistic",d.get("realistic_names",{}).get("ms_per_call_median"),d.get("realistic_names",{}).get("trie_rows_of_tokens"),"n1500",d.get("n1500",{}).get("ms_per_call_median"),"sdxl",d["sdxl"].get("ms_per_call_median"),"nsp",d["no_shared_prefix"].get("ms_per_call_median"),
              "lat1000",{k:round(v,2) for k,v in d.get("latency_n1000",{}).items() if k in("p50","p95","p99","max")},"lat100",{k:round(v,2) for k,v in d.get("latency_n100",{}).items() if k in("p50","p95","p99","max")})
PY
cat gpurun_out/r05_mfma16_ab2.txt
echo done
