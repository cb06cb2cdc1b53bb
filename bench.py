#!/usr/bin/env python3
"""bench.py — concept-edits/sec of the closed-form mass-edit path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run)

Workload (config.workload): the 1 000-concept batch on SD-v1.4 dims the metric is quoted on — CLIP ViT-L/14
text encoder (768 / 3072 / 12 layers, random init), layers [7, 8, 9, 10], lambda = 4000, edit_weight 0.5,
3 prompts per concept, synthetic v* and synthetic second moments C_l (no network for real weights/captions).
A STEP is one pass of the hot path over that batch with every input resident in HBM (token ids, lookup
indices, v*, C_l, encoder weights): restore the original fc2 weights, then edit_engine.run_encoder_edit —
one partial, prefix-deduplicated encoder forward (clip_forward.py) that runs gather -> assemble -> Cholesky ->
TRSM -> dW on the HIP kernels at each edited layer's fc2.  Host preparation (tokenizer, subject search, npz reads) happens once before the timed region and is
reported separately as `host_prepare_ms` (DESIGN.md §Measurement gives the all-inclusive rate).
With N > 1 the 1 000 concepts are sharded over the ranks (strong scaling, fixed total work): each rank
forwards its shard, K/Zc are all-gathered over RCCL per layer, every rank assembles and factors A, the triangular
solves and dW are split by concept rows and the partial U (h x d, fp64) is all-reduced.

`roofline`: after the timed region the same K steps run once more with every fp64-MFMA kernel class of the solve
bracketed by HIP events on the launch stream (emcid_profile_*); the class with the most time is reported with its
ALGORITHMIC flops per launch (SURVEY.md §8d counts); `kernel_classes` lists all of them.
`cpu_baseline`: the oracle (op-for-op CPU port of the reference path) timed on the host cores on a
100-concept sample of the same workload, rank 0, N == 1 only; the same run yields `dw_max_abs_err`.
"""
import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak [external: AMD MI355X datasheet; MI355X_MICROARCH.md lists no fp64 row]
LAYERS = (7, 8, 9, 10)
LAM, EW = 4000, 0.5
KIND = "sd-v1.4"


def build_inputs(n_concepts, device, workdir, shard=None):
    from emcid_amd import emcid_main as em, synthetic as syn
    from emcid_amd.emcid_hparams import EMCIDHyperParams

    pipe = syn.build_pipe(KIND, device, syllables=True)
    hidden, inter = syn.ENCODER_DIMS[KIND][:2]
    reqs = syn.make_requests(n_concepts, names="syllable")   # 3-token names, 7-token prompts like real CLIP BPE
    hp_d = syn.sd_hparams_dict(layers=LAYERS, mom2_update_weight=LAM, edit_weight=EW)
    cache = str(Path(workdir) / f"cache_{n_concepts}") + "/"
    stats = Path(workdir) / "stats"
    layer_names = [hp_d["rewrite_module_tmp"].format(l) for l in LAYERS]
    if not Path(cache).exists():
        syn.write_vstar_cache(cache, reqs, hidden, seed=1, scale=0.5)
    if not stats.exists():
        syn.write_stats_cache(stats, layer_names, inter, hp_d["mom2_n_samples"], seed=2, t=2 * inter)
    return pipe, reqs, hp_d, cache, str(stats), layer_names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--concepts", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch.distributed as dist
    from emcid_amd import emcid_main as em, hip
    from emcid_amd.edit_engine import ConceptShard, run_encoder_edit, check_info
    from emcid_amd.emcid_hparams import EMCIDHyperParams
    from emcid_amd.nethook import get_parameter

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
    backend = os.environ.get("EMCID_BENCH_BACKEND", "nccl")      # "gloo": functional test of this script on a 1-GPU box
    if backend == "gloo":
        local = 0                                                 # every rank shares cuda:0, collectives staged via host
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
    shard = ConceptShard(rank, world, None)

    # the benchmark models a long-running editing service: the projection GEMM solutions are tuned once per shape (in
    # prepare, untimed, reported as gemm_tuning_ms); a one-off library call only loads an existing results file
    os.environ.setdefault("EMCID_TUNE_GEMM", "1")
    workdir = Path(tempfile.gettempdir()) / f"emcid_bench_{os.getuid()}"
    if rank == 0:
        workdir.mkdir(exist_ok=True)
        build_inputs(args.concepts, "cpu", workdir)     # writes the synthetic v*/stats caches once
    if world > 1:
        dist.barrier()
    pipe, reqs, hp_d, cache, stats, layer_names = build_inputs(args.concepts, device, workdir)

    # ---- host preparation (outside the timed region): tokenizer, subject search, v*/C reads -> HBM ----------
    hp = EMCIDHyperParams(**hp_d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan = em.prepare_text_encoder_edit(pipe.text_encoder, pipe.tokenizer, reqs, hp, hp.layers, hp.mom2_update_weight,
                                        stats, cache, "", verbose=False, shard=shard)
    torch.cuda.synchronize()
    gemm_tuning_ms = plan.gemm_tuning_s * 1e3     # one-off: TunableOp picks the projection GEMM solutions (per shape, cached)
    host_prepare_ms = (time.perf_counter() - t0) * 1e3 - gemm_tuning_ms
    originals = {l: get_parameter(pipe.text_encoder, plan.weight_name(l)).detach().clone() for l in LAYERS}

    def step():
        with torch.no_grad():
            for l in LAYERS:
                get_parameter(pipe.text_encoder, plan.weight_name(l)).copy_(originals[l])
        return run_encoder_edit(plan, keep_factors=False, restore=False)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    check_info(plan)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    value = args.concepts * args.steps / elapsed

    # ---- roofline: the same K steps once more with every MFMA kernel class of the solve bracketed by HIP events
    # on the launch stream (emcid_profile_*; the graph replay is bypassed while events are recorded, the kernels and
    # their arguments are identical).  The class with the most time is reported. --------------------------------------
    mfma_classes = ["assemble", "chol_panel", "chol_trail", "chol_inner", "trsm_diag", "trsm_update", "delta_w", "inv_build",
                    "inv_apply", "inv_block"]
    # For this pass the covariance factorization runs on the SAME stream as everything else: with the two streams
    # overlapped, an event pair around a small kernel of one stream also counts the time it queued behind the other's.
    side = plan.side_stream
    plan.side_stream = torch.cuda.current_stream()
    hip.profile_enable(mfma_classes + ["chol_leaf"])
    for _ in range(args.steps):
        step()
    sync()
    prof = hip.profile_collect()
    hip.profile_enable([])
    plan.side_stream = side
    d, h = 3072, 768
    n_rows = (lambda b: b[1] - b[0])(shard.bounds(args.concepts)) if world > 1 else args.concepts
    N, L = args.concepts, len(LAYERS)
    dual = plan.dual_ws is not None
    Np = -(-N // 128) * 128

    def chol_flops(n):          # (leaf+panel excluded) trailing updates of one n x n Cholesky, SYRK count, two-level schedule
        nbk_, nob_ = n // 128, n // 512
        inner = sum(((n - (j + 1) * 128) * w - w * w // 2) * 128
                    for j in range(nbk_ - 1) for w in [(j // 4 + 1) * 512 - (j + 1) * 128] if w > 0)
        trail = sum(((n - J * 512) * 512 - 512 * 512 // 2) * (J * 512) for J in range(1, nob_))
        panel = sum((n - (j + 1) * 128) * 128 * 128 for j in range(nbk_ - 1))
        return {"chol_inner": inner, "chol_trail": trail, "chol_panel": panel}

    def trsm_flops(rows, n, directions=2):    # right-sided two-level TRSM of `rows` rows against an n x n factor
        nob_ = n // 512
        return {"trsm_update": directions * (2 * rows * 512 * sum(n - (J + 1) * 512 for J in range(nob_))),
                "trsm_diag": directions * (rows * nob_ * 512 * 512)}

    # flops per STEP and class, counted like SURVEY.md §8d (SYRK = n^2 k, triangular solve = rows * n^2 per direction)
    # for the launches each solver actually makes
    per_step_flops = {c: 0 for c in ("assemble", "chol_inner", "chol_trail", "chol_panel", "trsm_update", "trsm_diag", "delta_w",
                                     "inv_apply", "inv_build")}

    def add(table, times=1):
        for c, f in table.items():
            per_step_flops[c] += times * f

    if dual:
        # batched Cholesky of lam*C' (d x d) for the L layers + its explicit inverse factor X = inv(L) (d^3/3: block row I
        # costs 512*(512 I)^2 + 512^2*(512 I)); per layer (apply-only form): Yt = Kt X^T on the N concept rows (rows*d^2,
        # the triangular-solve count), SYRK S = I + Yt Yt^T, Cholesky of S, two solves of h rows against S's factor,
        # V = Z^T Yt, U = V X on h rows (h*d^2)
        add(chol_flops(d), L)
        add(chol_flops(Np), L)
        add(trsm_flops(h, Np, 2), L)
        # the first edited layer(s) substitute with L (forward on the concept rows, backward on h rows); the others
        # multiply by X = inv(L), which is built for them only (edit_engine: EMCID_INVERSE_FROM, default 1)
        first_x = min(L, max(0, int(os.environ.get("EMCID_INVERSE_FROM", "1"))))
        add(trsm_flops(n_rows, d, 1), first_x)
        add(trsm_flops(h, d, 1), first_x)
        per_step_flops["inv_apply"] = (L - first_x) * (n_rows + h) * d * d
        per_step_flops["inv_build"] = (L - first_x) * (d ** 3 // 3 - (d // 512) * 512 ** 3 // 3)   # minus the 512-blocks
    else:
        add(chol_flops(d), L)
        add(trsm_flops(n_rows, d, 2), L)
        per_step_flops["assemble"] = L * N * d * d
        per_step_flops["delta_w"] = L * 2 * h * n_rows * d
    classes = {c: {"ms_per_step": prof[c][0] / args.steps, "launches_per_step": prof[c][1] / args.steps}
               for c in prof}
    roofline = None
    cands = [c for c in per_step_flops if c in prof and prof[c][1] and per_step_flops[c] > 0]
    if cands:
        top = max(cands, key=lambda c: prof[c][0])
        ms, launches = prof[top]
        achieved = per_step_flops[top] * args.steps / (ms * 1e-3) / 1e12      # = flops per launch / average launch duration
        names = {"assemble": "gemm_f64_kernel<...> launched as SYRK (K^T K, or Pt Kt^T in the dual solver)",
                 "chol_trail": "gemm_f64_kernel<KC,KC,*,*,16,EpiAxpby> launched as left-looking Cholesky block-column update",
                 "chol_inner": "gemm_f64_kernel<KC,KC,*,64,16,2,2,EpiAxpby> launched as in-block Cholesky trailing update",
                 "chol_panel": "gemm_f64_kernel<KC,KC,32,64,16,2,2,EpiAxpby> launched as Cholesky panel solve",
                 "trsm_update": "gemm_f64_kernel<KC,*,*,*,16,EpiAxpby> launched as rank-512 TRSM update",
                 "trsm_diag": "gemm_f64_kernel<KC,*,32,64,16,2,2,EpiAxpby> launched as TRSM diagonal-block multiply",
                 "delta_w": "gemm_f64_kernel<!KC,*,64,64,16,2,2,EpiDeltaW> (dW = R^T X)",
                 "inv_apply": "gemm_f64_streamk_kernel<KC,*,128,128,16,2,4> (+ zero2d_f64_kernel of its output): GEMM against the "
                              "explicit inverse factor (Kt X^T, V X; triangular K range cut into 256 equal runs)",
                 "inv_build": "gemm_f64_kernel<KC,!KC,*,*,16,*> launched as recursive-halving build of X = inv(L)"}
        # HBM-side bytes per launch of that kernel come from separate `rocprofv3 --pmc` runs (FETCH_SIZE and WRITE_SIZE
        # cannot share a pass and neither can be collected from inside this process): scripts/pmc_inv_apply.py replays the
        # class's larger launch (1024 x 3072 x 3072); the summary, with the gfx950 FETCH_SIZE correction, is committed
        traffic, traffic_note = None, None
        pmc = Path(__file__).resolve().parent / "profiles" / "r01_pmc_inv_apply.json"
        if top == "inv_apply" and pmc.exists():
            with open(pmc) as f:
                rec = json.load(f)
            traffic = rec["traffic_bytes_per_launch"]
            traffic_note = (f"bytes per launch of the {rec['shape']['M']}x{rec['shape']['N']}x{rec['shape']['K']} launch, "
                            f"rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, profiles/{pmc.name}; algorithmic "
                            f"{rec['algorithmic_bytes_per_launch']} B")
        roofline = {"bound": "mfma", "kernel": names[top], "class": top, "achieved": achieved,
                    "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / F64_MFMA_PEAK_TFLOPS,
                    "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_note": traffic_note, "avg_launch_us": ms * 1e3 / launches, "launches": launches,
                    "flops_per_launch": per_step_flops[top] * args.steps / launches,
                    "solver": "dual" if dual else "direct"}

    out = {
        "metric": "concept-edits/sec (1 000-concept batch, SD-v1.4)", "value": value, "unit": "concept-edits/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.concepts}-concept edit, SD-v1.4 text-encoder dims (768/3072/12L), layers 7-10, "
                               f"lambda 4000, 3 prompts/concept, v* and C_l pre-cached in HBM",
                   "concepts": args.concepts, "prompts_per_rank": plan.batch.n_prompts,
                   "seq_len": int(plan.batch.inputs["input_ids"].shape[1]),
                   "forward": ("prefix-trie: %d unique rows of %d tokens" % (plan.trie.n_nodes, plan.trie.n_tokens_dense))
                   if plan.trie is not None else "hooked HF forward",
                   "parallelism": f"concept-shard x{world}"},
        "host_prepare_ms": host_prepare_ms,
        "gemm_tuning_ms": gemm_tuning_ms,
        "roofline": roofline,
        "kernel_classes": classes,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out.update(cpu_baseline_and_error(workdir, device))
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_and_error(workdir, device, n_sample=100):
    """Oracle on the host cores over a 100-concept sample of the workload + dW error of the HIP path on it."""
    import copy
    from emcid_amd import emcid_main as em, synthetic as syn
    from emcid_amd.emcid_hparams import EMCIDHyperParams
    from emcid_amd.nethook import get_parameter
    from oracle import emcid_oracle as orc

    cores = torch.get_num_threads()
    pipe_c, reqs, hp_d, cache, stats, layer_names = build_inputs(n_sample, "cpu", workdir)
    w0 = {ln: orc.get_parameter(pipe_c.text_encoder, ln + ".weight").clone() for ln in layer_names}
    t0 = time.perf_counter()
    orc.apply_emcid_to_text_encoder(pipe_c, reqs, copy.deepcopy(hp_d), cache_name=cache, stats_dir=stats)
    cpu_s = time.perf_counter() - t0
    pipe_g = syn.build_pipe(KIND, device, syllables=True)
    em.apply_emcid_to_text_encoder(pipe_g, reqs, EMCIDHyperParams(**hp_d), device, cache_name=cache, stats_dir=stats,
                                   verbose=False)
    err_abs, err_rel = 0.0, 0.0
    for ln in layer_names:
        ref = orc.get_parameter(pipe_c.text_encoder, ln + ".weight").double() - w0[ln].double()
        got = get_parameter(pipe_g.text_encoder, ln + ".weight").cpu().double() - w0[ln].double()
        e = (got - ref).abs().max().item()
        err_abs = max(err_abs, e)
        err_rel = max(err_rel, e / ref.abs().max().item())
    return {"cpu_baseline": {"value": n_sample / cpu_s, "unit": "concept-edits/s", "cores": cores, "kind": "port",
                             "sample": f"{n_sample}-concept edit (300 prompts), same dims/layers/lambda, one run of the "
                                       f"oracle's op-for-op port incl. its tokenization and npz reads ({cpu_s:.1f} s)"},
            "dw_max_abs_err": err_abs, "dw_max_rel_err": err_rel}


if __name__ == "__main__":
    main()
