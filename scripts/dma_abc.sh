out=gpurun_out/dma_abc; mkdir -p $out
for i in 1 2; do for v in 0 1 2; do
  EMCID_SP16_DMA=$v python bench.py --steps 60 --warmup 5 --no-stage0 --no-cpu-baseline --no-gemm-ab --no-variants > $out/v${v}_$i.json 2> $out/v${v}_$i.err
done; done
python - <<PY
import json
for i in (1,2):
    for v in (0,1,2):
        d=json.loads(open(f"gpurun_out/dma_abc/v{v}_{i}.json").read().strip().splitlines()[-1])
        pc=sorted(d["ms_per_call"])
        print("DMA",v,"run",i,"mean",round(d["ms_per_step"],3),"median",round(d["ms_per_call_median"],3),"p10",round(pc[6],2),"p90",round(pc[53],2),"device",round(d["device_ms_per_step"],3),"linear",round(d["kernel_classes"]["linear"]["ms_per_step"],3))
PY
