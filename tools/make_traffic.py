"""Build profiles/traffic.json from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/collect_profiles.sh.

    python tools/make_traffic.py gpurun_out/<tag>_pmc_FETCH_SIZE gpurun_out/<tag>_pmc_WRITE_SIZE <tag> [workload]

Per kernel: median counter value over its launches (KB), corrected as MI355X_MICROARCH.md's HBM section prescribes for
gfx950 -- FETCH_SIZE counts wide coalesced reads at one half (x2), WRITE_SIZE is taken as is.  For the raster kernel
only the raster-materialising launches (OUT = true template argument) are used."""
import csv, glob, json, os, statistics, sys

NAMES = {  # kernel-name substring -> bench.py's launch name
    "k_raster_fast<true, true, true, 32, true": "raster_down_iq", "k_raster_fast4<": "raster_down_iq",
    "k_raster_fast<true, true, true, 32, false": "down_walk_iq",
    "k_raster_tile<true, true>": "raster_down_iq_exact",
    "k_shift_iir": "shift_iir", "k_proj": "sync_proj", "k_beta": "sync_beta", "k_tail": "sync_beta+shift_iir",
    "k_down_fused<true, 0, 0": "down_fused_iq_exact", "k_down_fused<true, 2, 2": "down_fused_iq_sums",
    "k_down_fused<true, 2, 0": "down_fused_iq", "k_guard": "sync_guard",
    "k_down_fused<true, 3, 2": "down_fused_iq_sums", "k_down_fused<true, 3, 0": "down_fused_iq",
    "k_raster_shear<true>": "raster_sheared_iq", "k_raster_shear<false>": "raster_unsheared_iq",
    # k_seg1024<KIND, CPLX>: KIND 0 = getWelch accumulator, 1 = getWaterfall writer, 2 = row store (tsdr_fft_c2c, 1024-point rows)
    "k_seg1024<0": "welch_seg1024", "k_seg1024<1": "waterfall_seg1024", "k_seg1024<2": "fft_rows1024",
}


def collect(d, counter):
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            for sub, name in NAMES.items():
                if sub in r["Kernel_Name"]:
                    out.setdefault(name, []).append(float(r["Counter_Value"]))
                    break
    return {k: statistics.median(v) for k, v in out.items()}


def search_totals(d, counter):
    """HBM-side bytes of ONE configuration search: every FFT / autocorrelation / argmax launch of the run summed, divided
    by the number of searches (= k_argmax launches).  --quick runs only the headline workload's search."""
    tot, n = 0.0, 0
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            if "k_fft_" in k or "k_ac_" in k or "k_argmax" in k or "k_amax_publish" in k:
                tot += float(r["Counter_Value"])
            if "k_argmax" in k or "k_amax_publish" in k:
                n += 1
    return (tot / n if n else 0.0), n


def run_total(d, counter):
    """sum of `counter` over every kernel launch of one profiled run"""
    tot = 0.0
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            # torch's own fill kernels and the one-off set-up work of init_resampler (f64 filter synthesis) are not part of a call
            if r["Counter_Name"] == counter and "at::native" not in k and not any(
                    t in k for t in ("k_fft64_pass", "k_scale64", "k_blue64", "k_mul64", "k_window64", "k_altsign64", "k_herm_half")):
                tot += float(r["Counter_Value"])
    return tot


def spectra_main():
    """make_traffic.py --spectra <leg> <calls> <fetch dir> <write dir> <tag>: HBM-side bytes per call of one GetSpectrum /
    resampler leg from the PMC passes of `bench.py --spectra-only <leg> --steps <calls>` (every library kernel of the run
    summed -- torch's own fill kernels excluded -- and divided by the number of calls)."""
    leg, calls, fdir, wdir, tag = sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    try:
        doc = json.load(open(path))
    except Exception:
        doc = {}
    f, w = run_total(fdir, "FETCH_SIZE") / calls, run_total(wdir, "WRITE_SIZE") / calls
    doc.setdefault("spectra", {})[leg] = {"fetch_size_kb": round(f), "write_size_kb": round(w), "hbm_bytes_per_call": int(f * 1024 * 2 + w * 1024),
                                          "calls": calls, "tag": tag}
    json.dump(doc, open(path, "w"), indent=1)
    print(leg, json.dumps(doc["spectra"][leg]))


def stats_main():
    """make_traffic.py --stats <kernel_stats.csv> <tag> [workload]: the rocprofv3 --kernel-trace --stats averages of the frame-loop
    kernels into profiles/traffic.json (per kernel: "rocprofv3": {avg_launch_ms, calls, tag}), so that bench.py can print the
    profiler's average beside its own HIP-event one (roofline.profiled)."""
    path_csv, tag = sys.argv[2], sys.argv[3]
    wl = sys.argv[4] if len(sys.argv) > 4 else "C2"
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    try:
        doc = json.load(open(path))
    except Exception:
        doc = {}
    kb = {"raster_down_iq": 485359920, "raster_down_iq_exact": 485359920, "down_fused_iq_sums": 137599920}   # C2 algorithmic bytes per launch
    for r in csv.DictReader(open(path_csv)):
        for sub, name in NAMES.items():
            if sub in r["Name"]:
                e = doc.setdefault(wl, {}).setdefault(name, {})
                ms = float(r["AverageNs"]) * 1e-6
                e["rocprofv3"] = {"avg_launch_ms": round(ms, 5), "calls": int(r["Calls"]), "tag": tag,
                                  "source": f"profiles/{tag}_kernel_stats_bench.csv"}
                if wl == "C2" and name in kb:
                    e["rocprofv3"]["achieved_GBs"] = round(kb[name] / (ms * 1e-3) / 1e9, 1)
                    e["rocprofv3"]["frac"] = round(kb[name] / (ms * 1e-3) / 1e9 / 8000.0, 4)
                break
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps({k: v.get("rocprofv3") for k, v in doc.get(wl, {}).items() if isinstance(v, dict) and "rocprofv3" in v}, indent=1))


def main():
    if sys.argv[1] == "--spectra":
        return spectra_main()
    if sys.argv[1] == "--stats":
        return stats_main()
    fdir, wdir, tag = sys.argv[1:4]
    wl = sys.argv[4] if len(sys.argv) > 4 else "C2"
    f, w = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    try:
        doc = json.load(open(path))
    except Exception:
        doc = {}
    doc["_note"] = ("HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace "
                    "only) on `python3 bench.py --quick --steps 5 --warmup 2` (MI355X, ROCm 7.2); built by tools/make_traffic.py from "
                    f"the passes tagged {tag}.  Counter unit KB, median over a kernel's launches.  gfx950 correction per "
                    "MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts coalesced reads at one half, so fetch bytes = "
                    "FETCH_SIZE*1024*2; WRITE_SIZE taken as is.  Infinity-Cache hits are included in FETCH_SIZE.")
    doc.setdefault(wl, {})   # (entries this pass did not measure keep their own tag)
    # (doc["_calibration"], written from tools/calib_fetch.sh's table, is kept as it is: the x2 correction was verified at 4, 8
    # and 16 bytes per lane in round 4)
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        e = doc[wl].setdefault(k, {})
        e.update({"fetch_size_kb": round(fk), "write_size_kb": round(wk), "hbm_bytes_per_launch": int(fk * 1024 * 2 + wk * 1024), "tag": tag})
    fs, ns = search_totals(fdir, "FETCH_SIZE")
    ws, _ = search_totals(wdir, "WRITE_SIZE")
    if ns:
        doc[wl]["search"] = {"fetch_size_kb": round(fs), "write_size_kb": round(ws), "hbm_bytes_per_search": int(fs * 1024 * 2 + ws * 1024),
                             "searches": ns, "tag": tag, "note": "all FFT passes + argmax of one configuration search (fused loaders / epilogue: no other kernels)"}
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc[wl], indent=1))


if __name__ == "__main__":
    main()
