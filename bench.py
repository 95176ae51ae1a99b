#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: megapixels/s, JPEG q=80, 1920x1080 batch (configs[1]).

A step = one pass of the whole hot path (entropy decode -> pixel-domain transcode -> entropy encode -> file
assembly) over one batch of synthetic 1080p JPEGs whose bytes are already resident in HBM.  One process per
GPU; files shard per rank with no collective on the data path (weak scaling: every rank gets --batch files).
Prints ONE JSON line on rank 0.

The headline (`value`, `ms_per_step`, `roofline`, `phases`) is the library's DEFAULT profile = what libcaesium's `-q 80` runs
(compressor.rs:415,427 -> mozjpeg JCP_MAX_COMPRESSION): scan search + trellis quantisation + overshoot deringing.  Measured in the same run
and carried in the line: `scalar_profile` (CSH_PROFILE=scalar: the scan search over the scalar quantiser -- the pieces pinned by j0.JPG and
libjpeg-turbo) and `plain_profile` (stock script, whole files equal libjpeg-turbo's); a `summary` object at the FRONT of the line repeats
the three values; `roofline` for the dominant kernel with `traffic` from two rocprofv3 --pmc passes this script starts itself (FETCH_SIZE,
WRITE_SIZE; separate passes, no trace domain); `cpu_baseline` (the oracle on the host, same profile), `boundary` (cs_batch_compress from
host buffers), `cli_end_to_end` (the caesiumclt binary, files in -> files out), `other_configs`
(configs[2], configs[3], configs[4]) each with its own roofline and CPU lines.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MP_1080P = 1920 * 1080 / 1e6


def algorithmic_bytes(kernel, t, n):
    """ALGORITHMIC bytes one launch of `kernel` must move for the batch (DESIGN.md 'Roofline numerators'):
    per 1080p 4:2:0 image: coefficient planes 6 266 880 B (Y 4 177 920 + 2 x 1 044 480), chroma planes 2 x 522 240 B."""
    coef = t.coef_bytes            # all components, one direction
    y = coef * 2 // 3              # luma share at 4:2:0 (32640 of 48960 blocks)
    c = coef - y
    planes = c // 2                # u8 chroma planes (1 B/sample vs 2 B/coefficient)
    table = {
        "k_decode_prog+seq": t.in_bytes + coef, "k_refine_chains": t.in_bytes + coef,
        "k_dec_spec": t.in_bytes, "k_dec_relax0": t.in_bytes,
        "k_dec_write": t.in_bytes + coef,           # stream in, coefficient planes out (SURVEY 8d phase D)
        # phase E reads the planes ONCE (k_tokens: every AC scan of a component from one load of its blocks -- the scan search's 28 / 33
        # candidate scans of a stage included) and writes the files; tokens are this design's intermediate, not algorithmic bytes
        "k_tokens": coef, "k_pack": t.out_bytes, "scan_search_stage2": coef,
        # the list builder reads the planes once (what k_tokens used to do per stage); the list passes read lists, not planes: their
        # algorithmic input is still the component's coefficients (SURVEY 8d phase E), their output the files
        "k_nzlist": coef, "k_list_stats": coef, "k_list_pack": t.out_bytes,
        "k_xform_direct": 2 * y,
        "k_idct_plane": c + planes,
        "k_resample+k_plane_fdct": 3 * planes + c,
        "memset_coef": 2 * coef,
        # the trellis quantiser reads the retained DCT once and writes the coefficients once (J7 inside phase X)
        "k_trellis_ac": 2 * coef, "trellis_stats": coef,
    }
    return table.get(kernel)


# hipEvent kernel slot -> substring of the rocprofv3 kernel name
ROCPROF_NAME = {"k_dec_write": "k_dec_dense<2", "k_dec_spec": "k_dec_dense<0", "k_dec_relax0": "k_dec_dense<1", "k_dec_relax1_4": "k_dec_relax_list",
                "k_resample+k_plane_fdct": "k_resample_fdct_420", "unstuff": "k_unstuff_copy", "k_emit": "k_emit_data", "trellis_stats": "k_nzlist",
                "scan_search_stage2": "k_list_pack"}


def live_pmc(kernel, batch, profile, inputs=None, timeout=120):
    """HBM bytes of ONE launch of `kernel`, measured now: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass,
    MI355X_MICROARCH.md "rocprofv3 PMC slots"; no trace domain next to them) over one step of this same workload in a child process
    (`--pmc-child`).  Counter unit KiB; FETCH_SIZE doubled (the gfx950 note of the same guide: wide coalesced reads are tallied at half their
    bytes), WRITE_SIZE as is (it reads exactly the bytes of the coefficient-pool memset).  Returns (bytes per launch or None, detail dict)."""
    import csv
    import glob
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, {"error": "rocprofv3 not on PATH"}
    key = ROCPROF_NAME.get(kernel, kernel)
    detail = {"kernel_regex": key, "method": "rocprofv3 --pmc <counter> --kernel-include-regex, one step, child process of this run"}
    total = 0.0
    # the child gets its input files from this process (a pickle): making them there means a fork pool inside a profiled process -- one such child hung
    import pickle
    handoff = None
    if inputs:
        fd, handoff = tempfile.mkstemp(prefix="csh_pmc_in_", suffix=".pkl", dir="/tmp")
        with os.fdopen(fd, "wb") as f:
            pickle.dump(list(inputs[:64]), f)
    try:
        return _live_pmc_passes(exe, key, detail, batch, profile, handoff, timeout)
    finally:
        if handoff:
            try:
                os.unlink(handoff)
            except OSError:
                pass


def _live_pmc_passes(exe, key, detail, batch, profile, handoff, timeout):
    import csv
    import glob
    total = 0.0
    for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        d = tempfile.mkdtemp(prefix="csh_pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        if handoff:
            env["CSH_PMC_INPUTS"] = handoff
        if profile:
            env["CSH_PROFILE"] = profile
        else:
            env.pop("CSH_PROFILE", None)
        cmd = [exe, "--pmc", counter, "--kernel-include-regex", key.replace("<", "."), "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--batch", str(batch)]
        try:
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
            except subprocess.TimeoutExpired:   # once more (once per run): the passes are independent processes
                if detail.get("retried"):
                    raise
                detail.setdefault("retried", []).append(counter)
                shutil.rmtree(d, ignore_errors=True)
                os.makedirs(d, exist_ok=True)
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
            if r.returncode != 0:
                err = "\n".join(ln for ln in r.stderr.decode(errors="replace").splitlines() if not (ln[:1] in "WEI" and ln[1:5].isdigit()))   # the child's own words, not the profiler's log lines
                return None, dict(detail, error=f"{counter}: rocprofv3 exited {r.returncode}: ...{err[-400:]}")
            vals = []
            for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(fn)):
                    if r.get("Counter_Name") == counter and key in r.get("Kernel_Name", ""):
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, dict(detail, error=f"no {counter} rows for {key}")
            per_launch = sum(vals) / len(vals) * 1024.0 * scale
            detail[counter] = {"dispatches": len(vals), "bytes_per_launch": int(per_launch), "scale": scale}
            total += per_launch
        except Exception as e:   # a profiler problem must not take the headline down
            return None, dict(detail, error=f"{counter}: {type(e).__name__}: {str(e)[-200:]}")
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(total), detail


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def pillow_proxy(src):
    """libjpeg-turbo through Pillow doing the plain-profile job: decode without colour conversion, re-encode with the q80 table #3,
    4:2:0, progressive, optimised Huffman (what tests/test_oracle_jpeg.py pins the oracle to)"""
    import io

    from PIL import Image
    im = Image.open(io.BytesIO(src))
    im.draft("YCbCr", im.size)
    im.load()
    b = io.BytesIO()
    im.save(b, format="JPEG", qtables=[Q80_TABLE3, Q80_TABLE3], subsampling=2, progressive=True, optimize=True)
    return b.getvalue()


def pillow_png_proxy(src):
    """libpng through Pillow: decode, re-encode at zlib level 9 with adaptive filtering (BASELINE.md section 3's PNG proxy; not oxipng)"""
    import io

    from PIL import Image
    b = io.BytesIO()
    Image.open(io.BytesIO(src)).save(b, format="PNG", compress_level=9)
    return b.getvalue()


def pillow_webp_proxy(src):
    """libjpeg-turbo + Pillow's Lanczos + libwebp method 4 through Pillow: JPEG -> long edge 1500 -> WebP q85 (BASELINE.md section 3's proxy)"""
    import io

    from PIL import Image
    im = Image.open(io.BytesIO(src)).convert("RGB")
    w, h = im.size
    nw, nh = (1500, round(h * 1500 / w)) if w >= h else (round(w * 1500 / h), 1500)
    b = io.BytesIO()
    im.resize((nw, nh), Image.LANCZOS).save(b, format="WEBP", quality=85, method=4)
    return b.getvalue()


# mozjpeg base table #3 at libjpeg scale 40 (= -q 80), natural order (pinned by samples/j0.JPG's DQT at scale 98)
_BASE3 = [16, 16, 16, 18, 25, 37, 56, 85, 16, 17, 20, 27, 34, 40, 53, 75, 16, 20, 24, 31, 43, 62, 91, 135, 18, 27, 31, 40, 53, 74, 106, 156,
          25, 34, 43, 53, 69, 94, 131, 189, 37, 40, 62, 74, 94, 124, 169, 238, 56, 53, 91, 106, 131, 169, 226, 311, 85, 75, 135, 156, 189, 238, 311, 418]
Q80_TABLE3 = [max(1, min(32767, (v * 40 + 50) // 100)) for v in _BASE3]


def _one_input(i):
    from gen_synth import synth_jpeg
    return synth_jpeg(i)


def _one_png_1080(i):
    from gen_synth import synth_png
    return synth_png(i, 1920, 1080, "RGB")


def _one_webp_1080(i):
    import io

    from PIL import Image

    from gen_synth import synth_rgb
    b = io.BytesIO()
    Image.fromarray(synth_rgb(i), "RGB").save(b, format="WEBP", quality=90, method=2)
    return b.getvalue()


def pool_map(fn, items):
    items = list(items)
    if len(items) <= 4:
        return [fn(i) for i in items]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(len(items), os.cpu_count() or 1, 192)) as pool:
        return pool.map(fn, items, chunksize=1)


def make_inputs(first, count):
    """`count` distinct 1080p q92 4:2:0 baseline JPEGs (SURVEY 8d recipe, seeds first..), made on all host cores (~1.2 s of numpy each)"""
    return pool_map(_one_input, range(first, first + count))


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* as
    torch.distributed.run would set them), pass rank 0's JSON line through, fail if any rank fails."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"bench ranks exited with {rcs}")
    return None


def set_profile(profile):
    if profile:
        os.environ["CSH_PROFILE"] = profile
    else:
        os.environ.pop("CSH_PROFILE", None)


def profile_record(api, pkg, blobs, params, local, profile, steps, names, note, parity_idx):
    """the same batch under another CSH_PROFILE: value, per-kernel times, output bytes, and a byte check of a few files against the oracle
    restated for that profile"""
    import torch

    from _util import oracle_lossy
    before = os.environ.get("CSH_PROFILE")
    set_profile(profile)
    try:
        pb = api.batch(blobs, params, device=local)
        pb.run()
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        ptm = [pb.run() for _ in range(steps)]
        torch.cuda.synchronize()
        pdt = (time.perf_counter() - p0) / len(ptm)
        outs = pb.fetch()
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(len(parity_idx), os.cpu_count() or 1)) as ex:
            parity = all(ex.map(lambda i: outs[i] == oracle_lossy(blobs[i]), parity_idx))
        rec = {"value": round(ptm[-1].pixels / 1e6 / pdt, 1), "unit": "MP/s", "ms_per_step": round(pdt * 1e3, 3), "out_bytes": int(ptm[-1].out_bytes),
               "parity_spot_check": bool(parity), "parity_files_checked": len(parity_idx),
               "kernel_ms": {names[i]: round(sum(x.kernel_ms[i] for x in ptm) / len(ptm), 3) for i in range(len(names)) if names[i] and ptm[-1].kernel_ms[i] > 0.05},
               "note": note}
        del outs
        pb.close()
        return rec, ptm
    finally:
        set_profile(before)


def timed_threads(fn, items, cores):
    from concurrent.futures import ThreadPoolExecutor
    c0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(fn, items))
    return time.perf_counter() - c0


def run_cli(args_list, env=None):
    exe = os.path.join(ROOT, "caesium-clt_amd", "bin", "caesiumclt")
    c0 = time.perf_counter()
    r = subprocess.run([exe] + args_list, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    return time.perf_counter() - c0, r


def scratch_dir():
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    return tempfile.mkdtemp(prefix="csh_bench_", dir=base)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2048, help="1080p files per rank per step (2048 x ~21 MB of device pools = 44 GB of the 288 GB)")
    ap.add_argument("--unique", type=int, default=1000, help="distinct synthetic images per rank (cycled to --batch): SURVEY 8d's 1 000 unique images; generated on all host cores")
    ap.add_argument("--cpu-images", type=int, default=96, help="files timed through the single-thread CPU oracle in the headline's profile (rank 0, N=1; ~0.15 s each: a 10-20 s sample); the all-core and Pillow lines scale from it")
    ap.add_argument("--boundary-files", type=int, default=2048, help="files of the cs_batch_compress (host buffers in, host buffers out) measurement; 0 = skip")
    ap.add_argument("--cli-files", type=int, default=2048, help="files of the caesiumclt end-to-end measurement (files in -> files out); 0 = skip")
    ap.add_argument("--no-extras", action="store_true", help="skip the boundary / CPU / other-config records (profiling runs)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--boundary", action="store_true", help="measure the BOUNDARY across --gpus ranks instead of the resident-in-HBM path: one shared file list, every rank calls "
                    "cs_batch_compress from host buffers on its own device (strong scaling; /root/reference/src/compressor.rs:81-100's par_iter as device shards)")
    ap.add_argument("--boundary-total", type=int, default=10000, help="files of the shared list of --boundary (SURVEY 8d: 10 000)")
    ap.add_argument("--same-device", action="store_true", help="--boundary: every rank uses device 0 (two ranks on one GPU: the code path of N ranks without N GPUs)")
    args = ap.parse_args()

    if args.pmc_child:
        # one step of the same workload for the counter passes (live_pmc): no torch, no records
        from _util import package
        pkg = package()
        api = pkg.load()
        if os.environ.get("CSH_PMC_INPUTS"):
            import pickle
            with open(os.environ["CSH_PMC_INPUTS"], "rb") as f:
                uniq = pickle.load(f)
        else:
            uniq = [_one_input(i) for i in range(16)]   # (no process pool in here: this process runs under the profiler)
        b = api.batch([uniq[i % len(uniq)] for i in range(args.batch)], pkg.default_parameters(jpeg_quality=80))
        b.run()
        b.close()
        return None

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)   # python bench.py --gpus N by itself: one process per GPU, this process only relays rank 0's line
    if args.boundary:
        return boundary_main(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)

    from _util import package
    pkg = package()
    api = pkg.load()
    if api.device_count() < 1:
        raise SystemExit("no HIP device: libcaesium_hip has no CPU path")

    # config 2 inputs: Pillow/libjpeg-turbo q92 4:2:0 baseline JPEGs of the SURVEY 8d synthetic images
    nuniq = max(1, min(args.unique, args.batch))
    uniq = make_inputs(rank * nuniq, nuniq)
    blobs = [uniq[i % nuniq] for i in range(args.batch)]
    params = pkg.default_parameters(jpeg_quality=80)
    batch = api.batch(blobs, params, device=local)   # parse + upload: inputs now resident in HBM

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run()
    barrier()
    t0 = time.perf_counter()
    timings = [batch.run() for _ in range(args.steps)]
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    t = timings[-1]
    assert t.n_images == args.batch and t.n_failed == 0
    prof_env = os.environ.get("CSH_PROFILE") or "mozjpeg"
    PROFILE_NOTES = {
        "mozjpeg": "the library's default = what libcaesium's -q runs: mozjpeg JCP_MAX_COMPRESSION -- scan search (optimize_scans in mozjpeg's own order, pinned by samples/j0.JPG) + trellis "
                   "quantisation + overshoot deringing (device == oracle byte for byte; the oracle's trellis / deringing restated from recall of mozjpeg 4.1: parity with the real crate UNPINNED, "
                   "tests/golden/make_reference_goldens.sh is the recipe that pins it)",
        "scalar": "CSH_PROFILE=scalar: mozjpeg's scan search over the scalar quantiser -- the pieces pinned by samples/j0.JPG and libjpeg-turbo",
        "plain": "CSH_PROFILE=plain: jpeg_simple_progression, no scan search, scalar quantiser; whole files byte-identical to libjpeg-turbo (tests/test_oracle_jpeg.py)",
    }
    profile = PROFILE_NOTES.get(prof_env, prof_env)
    mp_per_step = t.pixels / 1e6 * world
    value = mp_per_step * args.steps / dt

    # parity on this very batch (outside the timed region): EVERY distinct image of the batch, device output == oracle byte for byte (the all-core
    # oracle does ~150 MP/s: 1000 files in ~14 s); the two side profiles check every 8th
    outs = batch.fetch()
    from _util import oracle_lossy
    all_idx = list(range(nuniq)) if (world == 1 and not args.no_extras) else sorted({(i * nuniq) // 32 for i in range(32)} | {nuniq - 1})
    parity_idx = sorted(set(range(0, nuniq, 8)) | {nuniq - 1})
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(len(all_idx), os.cpu_count() or 1)) as ex:   # (ctypes releases the GIL inside the oracle)
        parity = all(ex.map(lambda i: outs[i] == oracle_lossy(blobs[i]), all_idx))
    del outs
    batch.close()   # its pools go back to the block cache: the records below make batches of their own

    # strong scaling at the boundary, in this same invocation (every rank takes part); then, N > 1 only, the drop-in binary over the same list on N devices
    bstrong = cli_gpus = None
    if not args.no_extras and args.boundary_total > 0:
        def reduce_max_sum(mx, sums):
            if world == 1:
                return mx, sums
            tmax = torch.tensor([mx], device="cuda", dtype=torch.float64); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = torch.tensor(sums, device="cuda", dtype=torch.float64); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            return float(tmax.item()), [float(x) for x in tsum.tolist()]
        try:
            bstrong = boundary_strong_leg(api, pkg, args, rank, world, local, barrier, reduce_max_sum)
        except Exception as e:   # a sub-record must not take the headline down (nor leave the other ranks in a collective: they fail alike)
            bstrong = {"error": repr(e)[:200]}
        if world > 1:
            api.release_cached_memory()   # rank 0's caesiumclt is about to use every device
            barrier()
            if rank == 0:
                cli_gpus = cli_gpus_leg(blobs, min(args.boundary_total, 10000), world)
            barrier()

    out = None
    if rank == 0:
        names = api.kernel_names()
        kms = [sum(tm.kernel_ms[i] for tm in timings) / len(timings) for i in range(len(names))]
        lumps = {"scan_search_stage2", "memset_coef", "memset_enc", "trellis_stats"}   # several launches under one timing slot / not a kernel of ours
        dom = max((i for i in range(len(names)) if names[i] not in lumps), key=lambda i: kms[i])
        ab = algorithmic_bytes(names[dom], t, args.batch)
        roof = {"bound": "hbm", "kernel": names[dom], "avg_ms": round(kms[dom], 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
        if ab is not None:
            ach = ab / (kms[dom] * 1e-3) / 1e9
            roof.update({"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(ab)})
        else:
            roof.update({"achieved": None, "frac": None})
        extras = world == 1 and not args.no_extras
        if world == 1 and not args.no_pmc:
            api.release_cached_memory()   # the closed batch's pools sit in this process's block cache: the child needs the same 100 GB
            traffic, detail = live_pmc(names[dom], args.batch, prof_env, inputs=blobs)
            roof["traffic"] = traffic
            roof["traffic_detail"] = detail
            if traffic and ab:
                roof["traffic_over_algorithmic"] = round(traffic / ab, 3)
        cpu = cpu_all = cpu_pillow = boundary = plain = scalar = cli = cli10k = None
        # the three phases of the path against SURVEY 8d's algorithmic bytes (D: stream in + planes out, X: planes in + out, E: planes in + files out)
        ph = [sum(tm.phase_ms[i] for tm in timings) / len(timings) for i in range(8)]
        coefb = t.coef_bytes

        def phase(ms, nbytes):
            return {"ms": round(ms, 3), "algorithmic_bytes": int(nbytes), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                    "frac_of_8TBps": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None}
        # the quantiser's kernels (trellis statistics, k_trellis_ac, k_trellis_dc) are timed under phase 1 by the library (SURVEY 8a J7 sits inside
        # phase X); listed apart here so that X stays the transform pair the north-star's read-roofline target names
        i_q = [names.index(n) for n in ("trellis_stats", "k_trellis_ac", "k_trellis_dc")]
        q_ms = sum(kms[i] for i in i_q)
        phases = {"D_entropy_decode": phase(ph[0], t.in_bytes + coefb), "X_pixel_transcode": phase(ph[1] - q_ms, 2 * coefb),
                  "X_read_only": phase(ph[1] - q_ms, coefb), "Q_trellis_quantiser": phase(q_ms, 2 * coefb), "E_entropy_encode": phase(sum(ph[2:8]), coefb + t.out_bytes)}
        sub_steps = max(2, args.steps // 2)
        if extras and prof_env != "scalar":
            scalar, _ = profile_record(api, pkg, blobs, params, local, "scalar", sub_steps, names, PROFILE_NOTES["scalar"], parity_idx)
        if extras and prof_env != "plain":
            plain, _ = profile_record(api, pkg, blobs, params, local, "plain", sub_steps, names, PROFILE_NOTES["plain"], parity_idx)
        if extras and args.cpu_images > 0:
            # the oracle in the headline's profile (what libcaesium's -q does: trellis + deringing + scan search) on one host thread: a bounded sample
            n = args.cpu_images
            c0 = time.perf_counter()
            for i in range(n):
                oracle_lossy(blobs[i % len(blobs)])
            cdt = time.perf_counter() - c0
            cpu = {"value": round(n * MP_1080P / cdt, 2), "unit": "MP/s", "cores": 1, "kind": "port",
                   "sample": f"{n} of the same 1080p files through oracle/jpeg_oracle.c in the headline's profile ({prof_env}: decode + IDCT + FDCT + deringing + trellis quantisation "
                             f"+ scan search + progressive optimal-Huffman coding), 1 thread, {cdt:.1f} s"}
            # the same port on every host core (the reference's rayon par_iter shape), and the libjpeg-turbo proxy (Pillow: decode to YCbCr,
            # re-encode q80 4:2:0 progressive + optimised tables -- the plain profile the oracle is pinned to) on every core
            cores = os.cpu_count() or 1
            m = min(cores * 2, 64 * n)
            cdt = timed_threads(lambda i: len(oracle_lossy(blobs[i % len(blobs)])), range(m), cores)   # ctypes releases the GIL inside the oracle
            cpu_all = {"value": round(m * MP_1080P / cdt, 2), "unit": "MP/s", "cores": cores, "kind": "port", "sample": f"{m} files, {cores} threads, {cdt:.1f} s, profile {prof_env}"}
            m = min(cores * 4, 128 * n)
            cdt = timed_threads(lambda i: len(pillow_proxy(blobs[i % len(blobs)])), range(m), cores)
            cpu_pillow = {"value": round(m * MP_1080P / cdt, 2), "unit": "MP/s", "cores": cores, "kind": "libjpeg-turbo proxy (Pillow), not libcaesium",
                          "sample": f"{m} files, {cores} threads, {cdt:.1f} s"}
        other = None
        if extras:
            other = other_configs(api, pkg, blobs, local)
        prog_inputs = None
        if extras:
            prog_inputs = progressive_inputs(api, pkg, local)
        if extras and args.boundary_files > 0:
            # the boundary itself: cs_batch_compress, host buffers in -> host buffers out (marker parse, pinned upload, kernels, download);
            # PCIe and the host side are inside this number and never inside `value`
            nb = min(args.boundary_files, len(blobs))
            api.cs_batch_compress(blobs[:min(nb, 64)], params, device=local)     # warm the block caches as a long-running caller would have
            tm = []
            res = api.cs_batch_compress(blobs[:nb], params, device=local, timing=tm)
            ok = sum(1 for r in res if isinstance(r, bytes))
            boundary = {"entry": "cs_batch_compress", "files": nb, "ok": ok, "seconds": round(tm[0], 4), "files_per_s": round(nb / tm[0], 1),
                        "value": round(nb * MP_1080P / tm[0], 1), "unit": "MP/s", "note": "host buffers in and out, one call, second call of the process"}
        if extras and args.cli_files > 0:
            api.release_cached_memory()   # another process is about to use the device
            cli = cli_end_to_end(blobs, min(args.cli_files, 4096))
            if args.cli_files >= 2048 and args.boundary_total >= 10000:
                cli10k = cli_end_to_end(blobs, 10000)   # BASELINE configs[1]'s own count: the cold process amortises
        def short(rec):
            return None if rec is None else {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "parity": rec["parity_spot_check"]}
        summary = {"unit": "MP/s", prof_env + " (headline)": {"value": round(value, 1), "ms_per_step": round(dt / args.steps * 1e3, 3), "parity": bool(parity)},
                   "scalar": short(scalar), "plain": short(plain), "roofline_kernel": roof["kernel"], "roofline_frac": roof.get("frac"),
                   "X_read_only_frac": phases["X_read_only"]["frac_of_8TBps"], "E_ms": phases["E_entropy_encode"]["ms"]}
        if other:   # the other BASELINE configs in short (value, dominant kernel, its roofline fraction): what a tail of the line keeps of them
            summary["other_configs"] = {k.split(" ")[0]: ({"value": v.get("value"), "unit": v.get("unit"), "dominant_kernel": v.get("dominant_kernel"), "dominant_ms": v.get("dominant_ms"),
                                                          "frac": (v.get("roofline") or {}).get("frac")} if isinstance(v, dict) and "error" not in v else v) for k, v in other.items() if not k.startswith("_")}
        # ONE line; the bulky sub-records first, the contract's keys, roofline, cpu_baseline, phases and the summary at the END of the line (what a tail of it keeps)
        out = {
            "kernel_ms": {names[i]: round(kms[i], 3) for i in range(len(names)) if names[i] and kms[i] > 0.02},
            "other_configs": other, "progressive_inputs": prog_inputs,
            "scalar_profile": scalar, "plain_profile": plain, "cpu_baseline_all_cores": cpu_all, "cpu_proxy_pillow": cpu_pillow,
            "cli_end_to_end": cli, "cli_end_to_end_10k": cli10k, "cli_end_to_end_gpus": cli_gpus, "boundary": boundary, "boundary_strong": bstrong,
            "host": {"nproc": os.cpu_count(), "cpu": cpu_model()},
            "bytes": {"in": int(t.in_bytes), "out": int(t.out_bytes), "coef_one_way": int(t.coef_bytes)},
            "metric": "megapixels/sec JPEG q=80 1920x1080 batch", "value": round(value, 1), "unit": "MP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 1920x1080 q92 4:2:0 baseline JPEGs -> -q 80 progressive, inputs resident in HBM",
                       "files_per_gpu_per_step": args.batch, "unique_images": nuniq, "sharding": f"files/{world} ranks, no collective",
                       "profile": profile},
            "parity_spot_check": bool(parity), "parity_files_checked": len(all_idx),
            "device_ms_per_step": round(sum(tm.total_ms for tm in timings) / len(timings), 3),
            "cpu_baseline": cpu, "roofline": roof, "phases": phases, "summary": summary,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return out


def boundary_strong_leg(api, pkg, args, rank, world, local, barrier, reduce_max_sum):
    """the host-bound STRONG-scaling number inside the default invocation (so that a 1/2/4/8-GPU sweep of `bench.py --gpus N` carries it next to the
    resident weak-scaling value): one shared list of --boundary-total 1080p files (every rank makes the same list), rank r takes files r, r + N, ...
    (caesium-clt_amd/sharding.py) and calls cs_batch_compress on them from host buffers on its own device -- marker parse on the host's cores, pinned
    upload, kernels, download all inside the time; value = all files / the slowest rank's seconds.  No collective on the data path."""
    nuniq = max(1, min(args.unique, 256, args.boundary_total))
    uniq = make_inputs(0, nuniq)
    files = [uniq[i % nuniq] for i in range(args.boundary_total)]
    mine = files[rank::world]
    # whatever happens on this rank, it takes part in the barrier and in the reductions below (with ok = -1e9 as the sign of its failure): a rank that raised in
    # front of a collective would leave the others waiting in it
    err, dt, ok = None, 0.0, -1e9
    try:
        params = pkg.default_parameters(jpeg_quality=80)
        api.cs_batch_compress(mine[:64], params, device=local)
    except Exception as e:
        err = e
    barrier()
    if err is None:
        try:
            tm = []
            res = api.cs_batch_compress(mine, params, device=local, timing=tm)   # tm: the seconds inside the C call (the ctypes wrapper's copies in and out of Python are the harness)
            dt = tm[0]
            ok = float(sum(1 for r in res if isinstance(r, bytes)))
            del res
        except Exception as e:
            err = e
    dt_max, sums = reduce_max_sum(dt, [float(ok), float(len(mine))])
    if sums[0] < 0 or err is not None:
        return {"error": ("this rank: " + repr(err)[:160]) if err is not None else "another rank failed"}
    nfiles = int(sums[1])
    return {"entry": "cs_batch_compress from host buffers, one shared list, file i -> rank i mod N", "files": nfiles, "ok": int(sums[0]), "seconds_slowest_rank": round(dt_max, 4),
            "files_per_s": round(nfiles / dt_max, 1), "value": round(nfiles * MP_1080P / dt_max, 1), "unit": "MP/s", "scaling": "strong", "n_gpus": world,
            "parse_threads_per_rank": min(16, os.cpu_count() or 1),
            "note": "PCIe, host parsing and the download are inside this number (never inside `value`); N ranks use up to 16 N host cores and N upload streams"}


def cli_gpus_leg(blobs, n, gpus):
    """caesiumclt --gpus N over n 1080p files, files in -> files out, whole process (the reference tool's shape of a run on N devices of one node)"""
    d = scratch_dir()
    try:
        os.makedirs(os.path.join(d, "in"))
        for k in range(n):
            with open(os.path.join(d, "in", f"f{k:05d}.jpg"), "wb") as f:
                f.write(blobs[k % len(blobs)])
        runs = []
        for _ in range(2):
            shutil.rmtree(os.path.join(d, "out"), ignore_errors=True)
            secs, r = run_cli(["-q", "80", "--quiet", "--gpus", str(gpus), "-o", os.path.join(d, "out"), os.path.join(d, "in")])
            if r.returncode != 0:
                return {"error": f"caesiumclt exited {r.returncode}: {r.stderr.decode()[-200:]}"}
            runs.append(round(secs, 3))
        best = min(runs)
        return {"command": f"caesiumclt -q 80 --quiet --gpus {gpus} -o out/ in/", "files": n, "files_written": len(os.listdir(os.path.join(d, "out"))), "seconds": best,
                "seconds_each_run": runs, "files_per_s": round(n / best, 1), "value": round(n * MP_1080P / best, 1), "unit": "MP/s", "scaling": "strong", "n_gpus": gpus}
    except Exception as e:
        return {"error": str(e)[:200]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def boundary_main(args):
    """`--boundary`: where multi-GPU scaling can actually break -- the host side.  One shared list of --boundary-total 1080p files (every rank
    makes the same list: same seeds); rank r takes files r, r + N, r + 2N, ... (the per-file round-robin of caesium-clt_amd/sharding.py), and calls
    cs_batch_compress on them from host buffers on its own device: marker parse on the host's cores, pinned upload, kernels, download, all
    inside the time.  No collective touches the data; the barrier and the MAX over ranks of the elapsed time go over gloo (CPU), so that two
    ranks may share one device (--same-device).  Prints ONE JSON line on rank 0: total files / slowest rank's time = strong scaling."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    from _util import package
    pkg = package()
    api = pkg.load()
    if api.device_count() <= local:
        raise SystemExit(f"rank {rank}: no HIP device {local}")
    nuniq = max(1, min(args.unique, 256, args.boundary_total))
    uniq = make_inputs(0, nuniq)                                   # the SAME list on every rank
    files = [uniq[i % nuniq] for i in range(args.boundary_total)]
    mine = files[rank::world]
    params = pkg.default_parameters(jpeg_quality=80)
    api.cs_batch_compress(mine[:64], params, device=local)         # block caches and code objects warm, as in a long-running caller
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    res = api.cs_batch_compress(mine, params, device=local)
    dt = time.perf_counter() - t0
    ok = sum(1 for r in res if isinstance(r, bytes))
    out_bytes = sum(len(r) for r in res if isinstance(r, bytes))
    stats = torch.tensor([dt, float(ok), float(out_bytes), float(len(mine))], dtype=torch.float64)
    if world > 1:
        mx = stats.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, ok, out_bytes, nfiles = float(mx[0]), int(sm[1]), int(sm[2]), int(sm[3])
        dist.destroy_process_group()
    else:
        nfiles = len(mine)
    out = None
    if rank == 0:
        out = {"metric": "megapixels/sec JPEG q=80 1920x1080 batch, boundary (cs_batch_compress from host buffers)", "value": round(nfiles * MP_1080P / dt, 1), "unit": "MP/s",
               "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": round(dt * 1e3, 1), "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32",
               "data": "synthetic", "config": {"workload": f"configs[1] at the boundary: one list of {nfiles} synthetic 1080p q92 JPEGs ({nuniq} distinct), -q 80, host buffers in and out",
                                               "sharding": f"file i -> rank i mod {world}, no collective on the data path", "devices": "all ranks on device 0" if args.same_device else "rank r on device r"},
               "files": nfiles, "ok": ok, "out_bytes": out_bytes, "files_per_s": round(nfiles / dt, 1), "seconds_slowest_rank": round(dt, 3),
               "host": {"nproc": os.cpu_count(), "cpu": cpu_model(), "parse_threads_per_rank": min(16, os.cpu_count() or 1)},
               "note": "PCIe and the host side are inside this number; every rank parses on up to 16 host threads (pipeline.cpp batch_create) and uploads from its own pinned pool, so N ranks "
                       "use up to 16 N cores and N upload streams of the host"}
        print(json.dumps(out))
    return out


def cli_end_to_end(blobs, n):
    """the drop-in binary, files in -> files out: caesiumclt -q 80 over n 1080p files in a directory (tmpfs when there is one), the
    reference's own shape of a run (/root/reference/src/main.rs:43-113); wall clock of the whole process, second run of two"""
    d = scratch_dir()
    try:
        os.makedirs(os.path.join(d, "in"))
        for k in range(n):
            with open(os.path.join(d, "in", f"f{k:05d}.jpg"), "wb") as f:
                f.write(blobs[k % len(blobs)])
        runs, traces = [], []
        for _ in range(2):
            shutil.rmtree(os.path.join(d, "out"), ignore_errors=True)
            secs, r = run_cli(["-q", "80", "--quiet", "-o", os.path.join(d, "out"), os.path.join(d, "in")], env=dict(os.environ, CSH_TRACE="1"))
            if r.returncode != 0:
                return {"error": f"caesiumclt exited {r.returncode}: {r.stderr.decode()[-200:]}"}
            runs.append(round(secs, 3))
            traces.append([ln for ln in r.stderr.decode(errors="replace").splitlines() if ln.startswith("[cli]")][-2:])
        best = min(runs)
        nout = len(os.listdir(os.path.join(d, "out")))
        return {"command": "caesiumclt -q 80 --quiet -o out/ in/", "files": n, "files_written": nout, "seconds": best, "seconds_each_run": runs, "files_per_s": round(n / best, 1),
                "value": round(n * MP_1080P / best, 1), "unit": "MP/s", "where": os.path.dirname(d), "stages": traces,
                "note": "one process per run, as the reference tool: process start, HIP initialisation and first-launch code loading, directory scan, reads, parse, device pool "
                        "allocation, upload, kernels, download and writes all inside the number; the faster of two runs (a run right behind another process's exit waits for the "
                        "driver to hand its VRAM back: `stages` shows where the time went)"}
    except Exception as e:
        return {"error": str(e)[:200]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def progressive_inputs(api, pkg, local, files=512):
    """progressive JPEG inputs (caesium's own outputs are progressive): device time of the decode phase of 512 x 1080p files -- libjpeg's stock
    script (Pillow progressive=True: four AC refinement scans) and this library's own -q 80 output (mozjpeg's script).  DC first / AC first scans go through
    the self-synchronising decoder; AC refinement scans are parsed one wave per scan and applied one lane per block (k_decode_refine.hip).  Every file
    == the oracle's lossless transcode on a spot check."""
    try:
        from _util import oracle_lossless
        from gen_synth import synth_jpeg
        names = api.kernel_names()
        stock = [synth_jpeg(i, progressive=True) for i in range(8)]
        own = api.batch_compress([synth_jpeg(i) for i in range(8)], pkg.default_parameters(jpeg_quality=80), device=local)
        rec = {"files": files, "note": "lossless transcode (decode + re-encode), inputs resident in HBM; decode_ms = the kernel slots of phase D"}
        for label, uniq in (("stock_script", stock), ("own_q80_output", own)):
            b = api.batch([uniq[i % 8] for i in range(files)], pkg.default_parameters(jpeg_optimize=True), device=local)
            b.run()
            t = b.run()
            outs = b.fetch()
            nd = names.index("k_idct_plane")
            rec[label] = {"decode_ms": round(sum(t.kernel_ms[:nd]), 2), "total_ms": round(t.total_ms, 2), "n_seq_decoded": int(t.n_seq_decoded), "n_refine_chains": int(t.n_refine_chains),
                          "k_refine_chains_ms": round(t.kernel_ms[names.index("k_refine_chains")], 2), "in_bytes_per_file": int(t.in_bytes // files),
                          "parity": bool(outs[0] == oracle_lossless(uniq[0]) and outs[7] == oracle_lossless(uniq[7]))}
            b.close()
        return rec
    except Exception as e:   # a sub-record must not take the headline down
        return {"error": repr(e)}


def other_configs(api, pkg, blobs, local):
    """the other single-GPU configurations of BASELINE.json as sub-records (device time of the whole path, inputs resident in HBM), each with
    the roofline of its dominant kernel (SURVEY 8d's algorithmic bytes) and CPU lines timed on this host:
    configs[2] 3840x2160 RGB8 PNGs --lossless --png-opt-level 3; configs[3] the same 1080p JPEGs -> WebP q85 at long edge 1500;
    configs[4] a mixed JPEG / PNG / WebP tree through the caesiumclt binary (-R -S -q 80), files in -> files out"""
    other = {}
    cores = os.cpu_count() or 1
    try:
        from gen_synth import synth_png
        import multiprocessing as mp
        with mp.get_context("fork").Pool(4) as pool:
            pngs = pool.starmap(synth_png, [(100 + k, 3840, 2160, "RGB") for k in range(4)])
        # the CPU line of this row on one of these very 4K files: the oracle takes about a minute for it on one thread, so it runs on a host thread of its
        # own from here on (ctypes releases the GIL) and is collected at the end of other_configs, behind the device work of configs[3] and configs[4]
        import threading
        from _util import oracle_png
        cpu4k = {}
        def _cpu4k():
            c0 = time.perf_counter()
            try:
                cpu4k["out"] = len(oracle_png(pngs[0], 3)); cpu4k["s"] = time.perf_counter() - c0
            except Exception as e:
                cpu4k["error"] = repr(e)[:200]
        th4k = threading.Thread(target=_cpu4k, daemon=True); th4k.start()
        npng = 256   # SURVEY 8d's count for a latency-bound row: k_png_huff is one workgroup per zlib stream (~0.4 s for a 4K file, whatever the count): the batch must be wide (128 files: 629 MP/s, 256: 704, 512: 750)
        pp = pkg.default_parameters(png_optimize=True, png_optimization_level=3)
        warm = api.png_batch(pngs[:2], pp, device=local); warm.run(); warm.close()     # code objects and allocator warm; the timed batch is new
        pb = api.png_batch([pngs[k % 4] for k in range(npng)], pp, device=local)
        ptm = pb.run()                                                                  # its FIRST run: a second one would find the files already inflated
        pouts = pb.fetch()
        pn = api.png_kernel_names()
        pdom = max(range(len(pn)), key=lambda i: ptm.kernel_ms[i])
        raw_bytes = 2160 * (1 + 3840 * 3)   # filtered stream of one file: 24 885 360 B
        # algorithmic bytes per launch, per file: inflate R idat + W raw; a tokenizer / filter pass over the four trial streams of -o3: R 4 x raw
        per_file = {"k_png_inflate": 13_000_000 + raw_bytes, "k_png_hist": 4 * raw_bytes, "k_png_brute": 5 * raw_bytes, "k_png_scores": 5 * raw_bytes,
                    "k_png_emit": raw_bytes, "k_png_unfilter": 2 * raw_bytes, "k_png_filter5": 6 * raw_bytes}
        abp = per_file.get(pn[pdom], raw_bytes) * npng
        rec = {"files": npng, "value": round(npng * 3840 * 2160 / 1e6 / (ptm.total_ms / 1e3), 1), "unit": "MP/s", "device_ms": round(ptm.total_ms, 1),
               "in_bytes": sum(len(pngs[k % 4]) for k in range(npng)), "out_bytes": sum(len(o) for o in pouts if isinstance(o, bytes)),
               "dominant_kernel": pn[pdom], "dominant_ms": round(ptm.kernel_ms[pdom], 1),
               "kernel_ms": {pn[i]: round(ptm.kernel_ms[i], 1) for i in range(len(pn)) if pn[i] and ptm.kernel_ms[i] >= 0.05},
               "roofline": {"bound": "hbm", "kernel": pn[pdom], "avg_ms": round(ptm.kernel_ms[pdom], 1), "algorithmic_bytes": int(abp), "achieved": round(abp / (ptm.kernel_ms[pdom] * 1e-3) / 1e9, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(abp / (ptm.kernel_ms[pdom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                            "note": "latency-bound stream work (one workgroup per zlib stream / dependent match loads), not bandwidth-bound"},
               "note": "the k_png_inflate slot (k_png_huff + k_png_lz77) is one workgroup per zlib stream: a latency, the same for 16 or 500 files; the other kernels scale with the file count"}
        pb.close()
        # CPU lines: libpng / zlib level 9 through Pillow on every core (BASELINE.md section 3's proxy), and the oracle (1 thread, a 1080p file: the 4K one takes a minute)
        m = min(cores, 128)
        cdt = timed_threads(lambda i: len(pillow_png_proxy(pngs[i % 4])), range(m), cores)
        rec["cpu_proxy_pillow"] = {"value": round(m * 3840 * 2160 / 1e6 / cdt, 2), "unit": "MP/s", "cores": min(cores, m), "kind": "libpng + zlib-9 proxy (Pillow), not oxipng",
                                   "sample": f"{m} of the same 4K files, decode + re-encode compress_level 9, {cdt:.1f} s"}
        other["configs[2] 4K PNG --lossless -o3"] = rec
        other["_th4k"] = (th4k, cpu4k)
    except Exception as e:   # a sub-record must not take the headline down
        other["configs[2] 4K PNG --lossless -o3"] = {"error": str(e)[:200]}
    try:
        nweb = 1024   # a step of the macroblock loop holds at most seven macroblocks of a picture: the batch is what fills the chip
        wb = api.webp_batch([blobs[k % len(blobs)] for k in range(nweb)], pkg.default_parameters(webp_quality=85, width=1500), device=local)
        wb.run()
        wtm = wb.run()
        wn = api.webp_kernel_names()
        wdom = max(range(len(wn)), key=lambda i: wtm.kernel_ms[i])
        # Lanczos R 6 220 800 / W 3 798 000 per file; the VP8 tail reads the 1500 x 844 YUV 4:2:0 (1 899 000 B) and writes levels + the file
        per_file = {"resize": 6_220_800 + 3_798_000, "k_webp_yuv": 3_798_000 + 1_899_000, "k_vp8_analyse+segments+loop": 2 * 1_899_000 + 4_304_448, "k_webp_hdr+decisions+bool+assemble": 4_304_448 + int(wtm.out_bytes) // nweb}   # 4 304 448 B: the level records of 4 982 macroblocks
        abw = per_file.get(wn[wdom], 1_899_000) * nweb
        rec = {"files": nweb, "value": round(nweb * MP_1080P / (wtm.total_ms / 1e3), 1), "unit": "source MP/s", "device_ms": round(wtm.total_ms, 1), "out_bytes": int(wtm.out_bytes),
               "dominant_kernel": wn[wdom], "dominant_ms": round(wtm.kernel_ms[wdom], 1),
               "kernel_ms": {wn[i]: round(wtm.kernel_ms[i], 1) for i in range(len(wn)) if wn[i] and wtm.kernel_ms[i] >= 0.05},
               "roofline": {"bound": "hbm", "kernel": wn[wdom], "avg_ms": round(wtm.kernel_ms[wdom], 1), "algorithmic_bytes": int(abw), "achieved": round(abw / (wtm.kernel_ms[wdom] * 1e-3) / 1e9, 1),
                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(abw / (wtm.kernel_ms[wdom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                            "note": "libwebp's rate-distortion mode decision, one workgroup walking each picture: VALU-bound (about 9.5 k vector instructions per macroblock, DESIGN 8), not bandwidth-bound"}}
        wb.close()
        m = min(cores * 2, 512)
        cdt = timed_threads(lambda i: len(pillow_webp_proxy(blobs[i % len(blobs)])), range(m), cores)
        rec["cpu_proxy_pillow"] = {"value": round(m * MP_1080P / cdt, 2), "unit": "source MP/s", "cores": cores, "kind": "libjpeg-turbo + Pillow Lanczos + libwebp method 4 proxy, not libcaesium",
                                   "sample": f"{m} files, {cores} threads, {cdt:.1f} s"}
        from _util import oracle_jpeg_to_webp
        c0 = time.perf_counter()
        for i in range(2):
            oracle_jpeg_to_webp(blobs[i], 85, 1500, 0)
        cdt = time.perf_counter() - c0
        rec["cpu_baseline"] = {"value": round(2 * MP_1080P / cdt, 2), "unit": "source MP/s", "cores": 1, "kind": "port",
                               "sample": f"2 of the same files through the oracle (libjpeg decode, Lanczos3, oracle/vp8enc_oracle.c = libwebp's encoder restated), 1 thread, {cdt:.1f} s"}
        other["configs[3] JPEG -> WebP q85 long edge 1500"] = rec
    except Exception as e:
        other["configs[3] JPEG -> WebP q85 long edge 1500"] = {"error": str(e)[:200]}
    try:
        api.release_cached_memory()
        other["configs[4] mixed JPEG/PNG/WebP tree, caesiumclt -R -S -q 80"] = mixed_tree(blobs)
        if isinstance(other.get("_th4k"), tuple) and other["_th4k"][0].is_alive():   # (said, not hidden -- one of the host's cores is busy beside this wall-clock leg)
            other["configs[4] mixed JPEG/PNG/WebP tree, caesiumclt -R -S -q 80"]["host_contention"] = f"the one-thread 4K PNG oracle run of configs[2] was still going on one of {os.cpu_count()} cores"
    except Exception as e:
        other["configs[4] mixed JPEG/PNG/WebP tree, caesiumclt -R -S -q 80"] = {"error": str(e)[:200]}
    th = other.pop("_th4k", None)
    if th:
        th[0].join(timeout=240)
        rec = other.get("configs[2] 4K PNG --lossless -o3")
        if isinstance(rec, dict) and "s" in th[1]:
            rec["cpu_baseline"] = {"value": round(3840 * 2160 / 1e6 / th[1]["s"], 3), "unit": "MP/s", "cores": 1, "kind": "port",
                                   "sample": f"one of the batch's 3840x2160 RGB8 files through oracle/png_oracle.c at -o3 (4 trials), 1 thread, {th[1]['s']:.1f} s, {th[1]['out']} bytes out"}
        elif isinstance(rec, dict):
            rec["cpu_baseline"] = {"error": th[1].get("error", "the 4K oracle run did not finish in time")}
    return other

def mixed_tree(blobs, per_type=96):
    """configs[4] on ONE device (the driver shards it over 8 with --gpus; SURVEY 8d cfg 5): a directory tree of the cfg-2 JPEGs and 1080p PNGs
    and WebPs of the same recipe, 1:1:1, through the binary: caesiumclt -q 80 -R -S -o out tree/ -- files in -> files out, wall clock"""
    d = scratch_dir()
    try:
        pngs = pool_map(_one_png_1080, range(200, 200 + 8))
        webps = pool_map(_one_webp_1080, range(300, 300 + 8))
        layout = {"a": ("jpg", lambda k: blobs[k % len(blobs)]), os.path.join("b", "c"): ("png", lambda k: pngs[k % 8]), os.path.join("d", "e", "f"): ("webp", lambda k: webps[k % 8])}
        nbytes = 0
        for sub, (ext, get) in layout.items():
            os.makedirs(os.path.join(d, "tree", sub))
            for k in range(per_type):
                data = get(k)
                nbytes += len(data)
                with open(os.path.join(d, "tree", sub, f"m{k:04d}.{ext}"), "wb") as f:
                    f.write(data)
        secs, r = run_cli(["-q", "80", "-R", "-S", "--json", "-o", os.path.join(d, "out"), os.path.join(d, "tree")])
        if r.returncode != 0:
            return {"error": f"caesiumclt exited {r.returncode}: {r.stderr.decode()[-200:]}"}
        res = json.loads(r.stdout.decode())
        files = res.get("files", [])
        ok = sum(1 for f in files if str(f.get("status", "")).lower() == "success")
        by_status = {}
        for f in files:
            by_status[str(f.get("status"))] = by_status.get(str(f.get("status")), 0) + 1
        nout = sum(len(fs) for _, _, fs in os.walk(os.path.join(d, "out")))
        n = 3 * per_type
        return {"command": "caesiumclt -q 80 -R -S --json -o out/ tree/", "files": n, "per_type": per_type, "status": by_status, "success": ok, "files_written": nout,
                "in_bytes": nbytes, "seconds": round(secs, 3), "files_per_s": round(n / secs, 1), "value": round(n * MP_1080P / secs, 1), "unit": "MP/s",
                "note": "one device, first run of the process, everything inside (scan, reads, three codec rows, writes); PNG -q goes through the lossy PNG row, WebP inputs through the VP8 decoder"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
