"""Child of tests/test_multi_gpu.py: one process per GPU (torch.distributed.run), RCCL backend -- or, with
MGPU_SHARE_ONE_GPU=1, every rank on cuda:0 and the collectives staged through the host over gloo (the sharded product
kernels then run on a real GPU at world size > 1 even where only one GPU exists; only the transport differs).
Checks, at the launched world size, that
  * HipFrames (frames of ONE buffer sharded over the ranks, all-gather, replicated combine) returns bit for bit what a
    single tsdr_frames call returns on rank 0's GPU: sync indices, every frame, the IIR state -- on every rank;
  * HipSearch with the sharded route (segment + halo partial sums, ONE all-reduce, then the non-linear step) matches
    the single-GPU autocorrelation within 2e-4 dB with the same argmax, and the automatic route choice is consistent.
Exit status 0 = all ranks passed."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    share = os.environ.get("MGPU_SHARE_ONE_GPU") == "1"  # every rank on cuda:0, collectives over gloo (host-staged)
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if share:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from tempest_loader import load_package
    tsdr = load_package()
    import importlib
    synth = importlib.import_module("tempestsdr_jl_amd.synth")
    par = importlib.import_module("tempestsdr_jl_amd.parallel")
    ctx = tsdr.Context(local)
    ok = True
    # ---- frames: 2 MS/s 800x600@60 mode, 7 frames (ragged over the ranks) + tail
    Fs, x_t, y_t, fv, nfr = 2.0e6, 1056, 628, 60.0, 7
    S = synth.samples_per_frame(Fs, fv)
    iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 9)
    for precision in ("exact", "fast"):
        ctx.set_precision(precision)
        ref_state = np.zeros((600, 800), np.float32, order="F")
        ref = ctx.frames(tsdr.SyncXY(ctx, 600, 800), iq, S, y_t, x_t, np.float32(0.1), ref_state)
        t_iq = torch.from_numpy(iq.view(np.float32)).to(dev)
        state = torch.zeros(480000, dtype=torch.float32, device=dev)
        frames = torch.empty(nfr * 480000, dtype=torch.float32, device=dev)
        idx = torch.zeros(2 * nfr, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        hf = par.HipFrames(ctx, tsdr.SyncXY(ctx, 600, 800), dev, world, rank)
        assert hf.run(t_iq, iq.size, S, y_t, x_t, np.float32(0.1), state, frames, idx) == nfr
        good = (np.array_equal(idx.cpu().numpy().reshape(nfr, 2), ref["sync_idx"]) and
                np.array_equal(state.cpu().numpy().view(np.uint32), ref_state.ravel(order="F").view(np.uint32)) and
                all(np.array_equal(frames.cpu().numpy()[f * 480000:(f + 1) * 480000].view(np.uint32),
                                   ref["frames"][f].ravel(order="F").view(np.uint32)) for f in range(nfr)))
        if not good:
            print(f"rank {rank}: HipFrames[{precision}] at world {world} differs from the single-GPU loop", flush=True)
            ok = False
        # gather-to-root: only the rendering rank receives the frames and runs stage 2
        root = world - 1
        state.zero_(); frames.zero_(); idx.zero_()
        torch.cuda.synchronize()
        hg = par.HipFrames(ctx, tsdr.SyncXY(ctx, 600, 800), dev, world, rank, mode="gather_root", root=root)
        assert hg.run(t_iq, iq.size, S, y_t, x_t, np.float32(0.1), state, frames, idx) == nfr
        if rank == root:
            good = (np.array_equal(idx.cpu().numpy().reshape(nfr, 2), ref["sync_idx"]) and
                    np.array_equal(state.cpu().numpy().view(np.uint32), ref_state.ravel(order="F").view(np.uint32)) and
                    all(np.array_equal(frames.cpu().numpy()[f * 480000:(f + 1) * 480000].view(np.uint32),
                                       ref["frames"][f].ravel(order="F").view(np.uint32)) for f in range(nfr)))
            if not good:
                print(f"rank {rank}: HipFrames[{precision}, gather_root] at world {world} differs from the single-GPU loop", flush=True)
                ok = False
    ctx.set_precision("fast")
    # ---- getWelch of one capture, segments sharded, one all-reduce of 1024 floats
    Lw = (iq.size // 1024) * 1024
    want = ctx.getWelch(Fs, iq[:Lw])[1]
    got = par.HipWelch(ctx, dev, world, rank).run(t_iq, Lw).cpu().numpy()
    if not np.max(np.abs(got - want)) < 2e-4:   # dB
        print(f"rank {rank}: HipWelch at world {world}: {np.max(np.abs(got - want)):.3e} dB off the single-GPU getWelch", flush=True)
        ok = False
    # ---- search: n = 4 * n_lags so that the sharded transform IS smaller and the all-reduce route is the honest choice
    n, n_lags = 240_000, 60_000
    assert par.search_route(n, n_lags, world) in ("sharded", "replicated")
    assert par.search_route(2 * n_lags, n_lags, world) == "replicated"   # the reference's own window: halo dominates
    single = par.HipSearch(ctx, dev, 1, 0)
    res1, pos1, _ = single.run(t_iq, n, n_lags)
    for route in ("sharded", None):
        hs = par.HipSearch(ctx, dev, world, rank, route=route)
        res, pos, _ = hs.run(t_iq, n, n_lags)
        err = float(torch.max(torch.abs(res - res1)).item())
        if not (err < 2e-4 and pos == pos1):
            print(f"rank {rank}: HipSearch route={route} err {err:.3e} dB argmax {pos} vs {pos1}", flush=True)
            ok = False
    flag = torch.tensor([0 if ok else 1], dtype=torch.int32, device="cpu" if share else dev)
    dist.all_reduce(flag)
    dist.destroy_process_group()
    if rank == 0:
        print("mgpu_child:", "OK" if flag.item() == 0 else "FAILED", f"(world {world})", flush=True)
    sys.exit(0 if flag.item() == 0 else 1)


if __name__ == "__main__":
    main()
