"""Two ranks sharing the one GPU of the test box (gloo rendezvous, collectives staged through the host):
the concept-sharded edit must leave both ranks with the weights of the single-process edit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(tmp, n_req):
    from emcid_amd import synthetic as syn
    reqs = syn.make_requests(n_req, ragged=True)
    hp_d = syn.sd_hparams_dict(layers=(1, 2, 3, 4), mom2_update_weight=50, edit_weight=0.6, mom2_n_samples=1000)
    names = [hp_d["rewrite_module_tmp"].format(l) for l in hp_d["layers"]]
    cache = tmp + "/cache/"
    if not os.path.exists(cache):
        syn.write_vstar_cache(cache, reqs, 32, seed=1, scale=0.5)
        syn.write_stats_cache(tmp + "/stats", names, 128, 1000, seed=2, t=512)
    return reqs, hp_d, names, cache


def _worker(rank, world, port, tmp, n_req, solver):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      EMCID_SOLVER=solver)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emcid_amd import emcid_main as em, synthetic as syn
        from emcid_amd.emcid_hparams import EMCIDHyperParams
        from emcid_amd.nethook import get_parameter
        reqs, hp_d, names, cache = _setup(tmp, n_req)
        pipe = syn.build_pipe("toy", "cuda:0")
        em.apply_emcid_to_text_encoder(pipe, reqs, EMCIDHyperParams(**hp_d), "cuda:0", cache_name=cache,
                                       stats_dir=tmp + "/stats", verbose=False)     # shard picked up from the process group
        out = {n: get_parameter(pipe.text_encoder, n + ".weight").cpu().numpy() for n in names}
        np.savez(f"{tmp}/rank{rank}.npz", **out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_req,solver", [(7, "direct"), (8, "direct"), (7, "dual"), (8, "dual")])
def test_two_ranks_match_single_process(tmp_path, n_req, solver, monkeypatch):
    from emcid_amd import emcid_main as em, synthetic as syn
    from emcid_amd.emcid_hparams import EMCIDHyperParams
    from emcid_amd.nethook import get_parameter
    tmp = str(tmp_path)
    reqs, hp_d, names, cache = _setup(tmp, n_req)
    mp.spawn(_worker, args=(2, _free_port(), tmp, n_req, solver), nprocs=2, join=True)
    monkeypatch.setenv("EMCID_SOLVER", solver)
    em.clear_caches()
    pipe = syn.build_pipe("toy", "cuda:0")
    w0 = {n: get_parameter(pipe.text_encoder, n + ".weight").cpu().numpy().copy() for n in names}
    em.apply_emcid_to_text_encoder(pipe, reqs, EMCIDHyperParams(**hp_d), "cuda:0", cache_name=cache,
                                   stats_dir=tmp + "/stats", verbose=False)
    r0, r1 = np.load(f"{tmp}/rank0.npz"), np.load(f"{tmp}/rank1.npz")
    for n in names:
        single = get_parameter(pipe.text_encoder, n + ".weight").cpu().numpy()
        dw = single.astype(np.float64) - w0[n]
        # every rank finishes the layer from the same gathered rows: equal up to the summation order of the split-K
        # f64 atomics (a last-bit fp32 flip at most)
        assert np.abs(r0[n].astype(np.float64) - r1[n]).max() <= 1e-6 * np.abs(dw).max()
        err = np.abs(r0[n].astype(np.float64) - single).max()
        assert err <= 1e-5 * np.abs(dw).max(), (n, err)                  # different batch split in the fp32 forward


def _stage0_worker(rank, world, port, tmp):
    import json
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from emcid_amd import synthetic as syn
        from emcid_amd.layer_stats import layer_stats_text_encoder_multi
        pipe = syn.build_pipe("toy", "cuda:0")
        names = ["encoder.layers.1.mlp.fc2", "encoder.layers.3.mlp.fc2"]
        layer_stats_text_encoder_multi(pipe.text_encoder, pipe.tokenizer, names, tmp + "/stats_sharded", sample_size=300,
                                       batch_tokens=600, data_path=tmp + "/caps.json", progress=None,
                                       shard=(rank, world))
    finally:
        dist.destroy_process_group()


def test_stage0_caption_shards_sum_to_single_process(tmp_path):
    """Caption-sharded Stage 0 on two ranks (all-reduce of mom2/count) == the single-process statistic."""
    from emcid_amd import synthetic as syn
    from emcid_amd.layer_stats import layer_stats_text_encoder_multi, stats_filename
    tmp = str(tmp_path)
    syn.write_captions(tmp + "/caps.json", 400, seed=4)
    mp.spawn(_stage0_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
    pipe = syn.build_pipe("toy", "cuda:0")
    names = ["encoder.layers.1.mlp.fc2", "encoder.layers.3.mlp.fc2"]
    single = layer_stats_text_encoder_multi(pipe.text_encoder, pipe.tokenizer, names, tmp + "/stats_single", sample_size=300,
                                            batch_tokens=600, data_path=tmp + "/caps.json", progress=None)
    for n in names:
        f = stats_filename(tmp + "/stats_sharded", "text_encoder", "ccs_filtered", n, "float32", ["mom2"], 600, 300)
        with np.load(f) as z:
            assert int(z["mom2.count"]) == single[n].mom2.count
            ref = single[n].mom2.mom2.numpy()
            assert np.abs(z["mom2.mom2"] - ref).max() <= 2e-5 * np.abs(ref).max()
