"""One process, SEVERAL devices through tsdr_group_* (RCCL inside the library): results against the single-context ones and
timings, as ONE JSON line.  bench.py runs this as a child process (own HIP / RCCL state, a timeout) when the box shows more
than one GPU; by hand:   python tools/group_devices.py 0,1,2,3 [workload]
A device listed more than once (0,0) exchanges by copies and adds instead of RCCL -- the split logic on a 1-GPU box.
A third argument sets the group's own switches: threads=0|1|2,pin=0|1 (tsdr_group_set_option "member_threads" / "pin_host")."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tempest_loader import load_package
T = load_package()
import importlib
synth = importlib.import_module("tempestsdr_jl_amd.synth")

devices = [int(d) for d in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
wl = sys.argv[2] if len(sys.argv) > 2 else "C2"
w = synth.WORKLOADS[wl]
Fs, x_t, y_t, fv = w["Fs"], w["x_t"], w["y_t"], w["fv"]
S = synth.samples_per_frame(Fs, fv)
nfr = int(round(w["acquisition"] * Fs)) // S
iq = synth.synth_leak(Fs, x_t, y_t, fv, S * nfr + 17)
out = {"devices": devices, "workload": wl, "frames_per_buffer": nfr, "rccl": len(set(devices)) == len(devices)}

def stage(what):   # (bench.py reports the last one reached when it has to end this process)
    print(f"[group_devices] {what}", file=sys.stderr, flush=True)


stage("creating the contexts and communicators")
ctx = T.Context(devices[0])
g = T.Group(devices)
gopts = dict(kv.split("=") for kv in sys.argv[3].split(",")) if len(sys.argv) > 3 else {}
if "threads" in gopts:
    g.set_option("member_threads", int(gopts["threads"]))
if "pin" in gopts:
    g.set_option("pin_host", int(gopts["pin"]))
out["member_threads"], out["pin_host"] = int(gopts.get("threads", 1)), int(gopts.get("pin", 0))
try:
    stage("frames: parity")
    # ---- frames: bit for bit the single-context result, both precisions, two successive buffers (lagged s_y, IIR state)
    same = True
    for prec in ("fast", "exact"):
        ctx.set_precision(prec); g.set_precision(prec)
        g.set_option("sync_guard_auto", 0); g.sync_reset()
        sync = T.SyncXY(ctx, 600, 800)
        s1 = np.zeros((600, 800), np.float32, order="F"); s2 = np.zeros((600, 800), np.float32, order="F")
        for part in (iq, iq[3 * S + 5:]):
            a = ctx.frames(sync, part, S, y_t, x_t, np.float32(0.1), s1)
            b = g.frames(part, S, y_t, x_t, np.float32(0.1), s2)
            same &= a["n_frames"] == b["n_frames"] and np.array_equal(a["sync_idx"], b["sync_idx"])
            same &= all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a["frames"], b["frames"]))
            same &= np.array_equal(s1.view(np.uint32), s2.view(np.uint32))
        sync.close()
    out["frames_bit_identical_to_single_context"] = bool(same)
    ctx.set_precision("fast"); g.set_precision("fast"); g.sync_reset()
    stage("frames: timing")
    sync = T.SyncXY(ctx, 600, 800)
    for name, f in (("single_context", lambda: ctx.frames(sync, iq, S, y_t, x_t, np.float32(0.1), s1)),
                    ("group", lambda: g.frames(iq, S, y_t, x_t, np.float32(0.1), s2))):
        f()
        t0 = time.perf_counter()
        for _ in range(3):
            f()
        out[f"frames_ms_per_buffer_from_host_memory_{name}"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
    r, ms = g.timing()
    out["frames_stage_ms_on_root_stream"] = [round(m, 4) for m in ms]
    sync.close()
    # ---- search: the reference's window (n = 2 indexMax) through both routes
    stage("search")
    G0, p0, _ = ctx.autocorr_search(iq, Fs, 0.0, 0.1, 50, 90)
    srch = {}
    for route in ("root", "sharded"):
        g.autocorr_search(iq, Fs, 0.0, 0.1, 50, 90, route=route)
        t0 = time.perf_counter()
        for _ in range(3):
            Gg, pg, _ = g.autocorr_search(iq, Fs, 0.0, 0.1, 50, 90, route=route)
        dt = (time.perf_counter() - t0) / 3
        r, ms = g.timing()
        srch[route] = {"ms_per_search_from_host_memory": round(dt * 1e3, 3), "route_taken": r,
                       "stage_ms_on_root_stream": [round(m, 4) for m in ms], "same_argmax_as_single_context": bool(pg == p0),
                       "max_abs_dB_diff_vs_single_context": float(np.max(np.abs(Gg - G0)))}
    out["search"] = srch
    out["all_reduce_bytes"] = 4 * int(round(0.1 * Fs))
    stage("getWelch")
    _, y1 = ctx.getWelch(Fs, iq)
    _, y2 = g.getWelch(Fs, iq)
    out["welch_max_abs_dB_diff_vs_single_context"] = float(np.max(np.abs(y1 - y2)))
finally:
    stage("closing")
    g.close()
    ctx.close()
print(json.dumps(out), flush=True)
