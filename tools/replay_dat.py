#!/usr/bin/env python3
"""Replay a recorded .dat capture through the GPU path (BASELINE config 1's flow; see tempestsdr.jl_amd/replay.py):

    python tools/replay_dat.py capture.dat --fs 20e6 [--format single|short|double] [--offset 420000] [--save out.npz]
    python tools/replay_dat.py --synthetic C2 --fs 20e6     # write a synthetic leak with writeComplexBinary first

Runs on cuda:0 through libtempest_hip.so; there is no CPU mode here (tests/test_replay.py runs the same flow on the
CPU oracle)."""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path", nargs="?")
    ap.add_argument("--fs", type=float, default=20e6)
    ap.add_argument("--format", default="single", choices=["single", "short", "double"])
    ap.add_argument("--offset", type=int, default=420_000)
    ap.add_argument("--line-method", default="gui", choices=["gui", "script"])
    ap.add_argument("--synthetic", help="workload name of synth.WORKLOADS to generate, write to a temporary .dat and replay")
    ap.add_argument("--save", help="write the aligned image and the autocorrelation to this .npz")
    args = ap.parse_args()
    from tempest_loader import load_package
    tsdr = load_package()
    import importlib
    replay = importlib.import_module("tempestsdr_jl_amd.replay")
    dat = importlib.import_module("tempestsdr_jl_amd.dat_files")
    path = args.path
    if args.synthetic:
        synth = importlib.import_module("tempestsdr_jl_amd.synth")
        wl = synth.WORKLOADS[args.synthetic]
        n = int(0.25 * wl["Fs"]) + args.offset
        iq = synth.synth_leak(wl["Fs"], wl["x_t"], wl["y_t"], wl["fv"], n)
        path = os.path.join(tempfile.mkdtemp(), f"dumpIQ_{args.synthetic}.dat")
        dat.writeComplexBinary(iq, path, "single")
        args.fs, args.format = wl["Fs"], "single"
        print(f"wrote {n} synthetic samples of {args.synthetic} to {path}")
    if not path:
        ap.error("give a .dat path or --synthetic")
    ctx = tsdr.Context(0)
    r = replay.replay_file(ctx, path, args.fs, args.format, offset=args.offset, line_method=args.line_method)
    print(f"refresh {r['fv']:.2f} Hz (peak at {r['fv_unrounded']:.4f}), line count {r['y_t']:.1f} (lag {r['lag']}), "
          f"mode {r['name']!r} -> {r['mode'].width}x{r['mode'].height}")
    print(f"vsync (s_y, s_x) = {r['sync']}, tau = {r['tau']} pixels, frame start {r['sample_offset']} samples after the offset")
    if args.save:
        np.savez_compressed(args.save, aligned=r["aligned"], image=r["image"], G=r["G"])


if __name__ == "__main__":
    main()
