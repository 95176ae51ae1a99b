#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: megapixels/s, JPEG q=80, 1920x1080 batch (configs[1]).

A step = one pass of the whole hot path (entropy decode -> pixel-domain transcode -> entropy encode -> file
assembly) over one batch of synthetic 1080p JPEGs whose bytes are already resident in HBM.  One process per
GPU; files shard per rank with no collective on the data path (weak scaling: every rank gets --batch files).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes(kernel, t, n):
    """ALGORITHMIC bytes one launch of `kernel` must move for the batch (DESIGN.md 'Roofline numerators'):
    per 1080p 4:2:0 image: coefficient planes 6 266 880 B (Y 4 177 920 + 2 x 1 044 480), chroma planes 2 x 522 240 B."""
    coef = t.coef_bytes            # all components, one direction
    y = coef * 2 // 3              # luma share at 4:2:0 (32640 of 48960 blocks)
    c = coef - y
    planes = c // 2                # u8 chroma planes (1 B/sample vs 2 B/coefficient)
    table = {
        "k_decode_seq": t.in_bytes + coef,
        "k_dec_spec": t.in_bytes, "k_dec_relax0": t.in_bytes,
        "k_dec_write": t.in_bytes + coef,           # stream in, coefficient planes out (SURVEY 8d phase D)
        # phase E reads the planes ONCE (k_tokens: every AC scan of a component from one load of its blocks -- the scan search's 28 / 33
        # candidate scans of a stage included) and writes the files; tokens are this design's intermediate, not algorithmic bytes
        "k_tokens": coef, "k_pack": t.out_bytes, "scan_search_stage2": coef,
        "k_xform_direct": 2 * y,
        "k_idct_plane": c + planes,
        "k_resample+k_plane_fdct": 3 * planes + c,
        "memset_coef": 2 * coef,
    }
    return table.get(kernel)


# hipEvent kernel slot -> substring of the rocprofv3 kernel name (profiles/*.csv)
ROCPROF_NAME = {"k_dec_write": "k_dec_dense<2", "k_dec_spec": "k_dec_dense<0", "k_dec_relax0": "k_dec_dense<1", "k_dec_relax1_4": "k_dec_relax_list",
                "k_resample+k_plane_fdct": "k_resample_plane", "unstuff": "k_unstuff_copy", "k_emit": "k_emit_data"}


def pmc_traffic(kernel, batch):
    """HBM bytes of ONE launch of `kernel` from the committed PMC passes (profiles/r02_pmc_{FETCH,WRITE}_SIZE_batch<B>.csv:
    separate rocprofv3 --pmc runs of this same command at the same --batch, --steps 1; raw counter unit KiB; FETCH_SIZE doubled
    per the gfx950 note in MI355X_MICROARCH.md, WRITE_SIZE as is -- it reads exactly 2*coef bytes on the pool memset).
    None when the batch differs from the profiled one or the files are absent."""
    import csv
    key = ROCPROF_NAME.get(kernel, kernel)
    tot = 0.0
    for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        path = os.path.join(ROOT, "profiles", f"r02_pmc_{counter}_batch{batch}.csv")
        if not os.path.exists(path):
            return None
        hit = [r for r in csv.reader(open(path)) if len(r) == 3 and key in r[0]]
        if not hit:
            return None
        tot += sum(float(r[2]) / max(1, int(r[1])) for r in hit) * 1024.0 * scale
    return int(tot)


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


# mozjpeg base table #3 at libjpeg scale 40 (= -q 80), natural order (pinned by samples/j0.JPG's DQT at scale 98)
_BASE3 = [16, 16, 16, 18, 25, 37, 56, 85, 16, 17, 20, 27, 34, 40, 53, 75, 16, 20, 24, 31, 43, 62, 91, 135, 18, 27, 31, 40, 53, 74, 106, 156,
          25, 34, 43, 53, 69, 94, 131, 189, 37, 40, 62, 74, 94, 124, 169, 238, 56, 53, 91, 106, 131, 169, 226, 311, 85, 75, 135, 156, 189, 238, 311, 418]
Q80_TABLE3 = [max(1, min(32767, (v * 40 + 50) // 100)) for v in _BASE3]


def _one_input(i):
    from gen_synth import synth_jpeg
    return synth_jpeg(i)


def make_inputs(first, count):
    """`count` distinct 1080p q92 4:2:0 baseline JPEGs (SURVEY 8d recipe, seeds first..), made on all host cores (~1.2 s of numpy each)"""
    if count <= 4:
        return [_one_input(first + i) for i in range(count)]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(count, os.cpu_count() or 1, 64)) as pool:
        return pool.map(_one_input, range(first, first + count), chunksize=1)


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* as
    torch.distributed.run would set them), pass rank 0's JSON line through, fail if any rank fails."""
    import socket
    import subprocess
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2048, help="1080p files per rank per step (2048 x ~21 MB of device pools = 44 GB of the 288 GB)")
    ap.add_argument("--unique", type=int, default=256, help="distinct synthetic images per rank (cycled to --batch); generated on all host cores")
    ap.add_argument("--cpu-images", type=int, default=64, help="files timed through the single-thread CPU oracle (rank 0, N=1); the all-core and Pillow lines scale from it")
    ap.add_argument("--boundary-files", type=int, default=512, help="files of the cs_batch_compress (host buffers in, host buffers out) measurement; 0 = skip")
    ap.add_argument("--no-extras", action="store_true", help="skip the boundary / CPU / other-config records (profiling runs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)   # python bench.py --gpus N by itself: one process per GPU, this process only relays rank 0's line

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
    from gen_synth import synth_jpeg
    pkg = package()
    api = pkg.load()
    if api.device_count() < 1:
        raise SystemExit("no HIP device: libcaesium_hip has no CPU path")

    # config 2 inputs: Pillow/libjpeg-turbo q92 4:2:0 baseline JPEGs of the SURVEY 8d synthetic images
    uniq = make_inputs(rank * args.unique, args.unique)
    blobs = [uniq[i % args.unique] for i in range(args.batch)]
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
    profile = "plain (stock jpeg_simple_progression script)" if os.environ.get("CSH_PROFILE") == "plain" else \
        "mozjpeg scan search (optimize_scans: 64 candidate scans coded per file, pinned by samples/j0.JPG)"
    mp_per_step = t.pixels / 1e6 * world
    value = mp_per_step * args.steps / dt

    # spot-check parity on this very batch (outside the timed region)
    outs = batch.fetch()
    from _util import oracle_lossy
    parity = all(outs[i] == oracle_lossy(blobs[i]) for i in range(min(2, args.unique)))
    del outs
    batch.close()   # its pools go back to the block cache: the records below make batches of their own

    out = None
    if rank == 0:
        names = api.kernel_names()
        kms = [sum(tm.kernel_ms[i] for tm in timings) / len(timings) for i in range(len(names))]
        lumps = {"scan_search_stage2", "memset_coef", "memset_enc"}   # several launches under one timing slot / not a kernel of ours
        dom = max((i for i in range(len(names)) if names[i] not in lumps), key=lambda i: kms[i])
        ab = algorithmic_bytes(names[dom], t, args.batch)
        roof = {"bound": "hbm", "kernel": names[dom], "avg_ms": round(kms[dom], 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": pmc_traffic(names[dom], args.batch)}
        if ab is not None:
            ach = ab / (kms[dom] * 1e-3) / 1e9
            roof.update({"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(ab)})
        if names[dom] == "k_tokens":
            roof["note"] = ("integer bit work, bound by instruction issue, not by HBM: 1.04 M waves of 4.3 k issue slots per 1024 files, stalled 61 % of their life at 3 waves per SIMD (profiles/r02_pmc_sq_*_batch1024.txt); the scan search launches it twice per step (stage 1 timed here, stage 2 inside scan_search_stage2)")
        else:
            roof.update({"achieved": None, "frac": None})
        cpu = cpu_all = cpu_pillow = boundary = plain = None
        extras = world == 1 and not args.no_extras
        # the three phases of the path against SURVEY 8d's algorithmic bytes (D: stream in + planes out, X: planes in + out, E: planes in + files out)
        ph = [sum(tm.phase_ms[i] for tm in timings) / len(timings) for i in range(8)]
        coefb = t.coef_bytes
        def phase(ms, nbytes):
            return {"ms": round(ms, 3), "algorithmic_bytes": int(nbytes), "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                    "frac_of_8TBps": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None}
        phases = {"D_entropy_decode": phase(ph[0], t.in_bytes + coefb), "X_pixel_transcode": phase(ph[1], 2 * coefb),
                  "X_read_only": phase(ph[1], coefb), "E_entropy_encode": phase(sum(ph[2:8]), coefb + t.out_bytes)}
        if extras and os.environ.get("CSH_PROFILE") != "plain":
            # the same batch under the plain profile (stock 10-scan script: the output that is byte-identical to libjpeg-turbo's)
            os.environ["CSH_PROFILE"] = "plain"
            try:
                pb = api.batch(blobs, params, device=local)
                pb.run()
                torch.cuda.synchronize()
                p0 = time.perf_counter()
                ptm = [pb.run() for _ in range(max(2, args.steps // 2))]
                torch.cuda.synchronize()
                pdt = (time.perf_counter() - p0) / len(ptm)
                plain = {"value": round(t.pixels / 1e6 / pdt, 1), "unit": "MP/s", "ms_per_step": round(pdt * 1e3, 3), "out_bytes": int(ptm[-1].out_bytes),
                         "kernel_ms": {names[i]: round(sum(x.kernel_ms[i] for x in ptm) / len(ptm), 3) for i in range(len(names)) if names[i] and ptm[-1].kernel_ms[i] > 0.05},
                         "note": "CSH_PROFILE=plain: jpeg_simple_progression, no scan search; byte-identical to libjpeg-turbo (tests/test_oracle_jpeg.py)"}
                pb.close()
            finally:
                del os.environ["CSH_PROFILE"]
        if extras and args.cpu_images > 0:
            n = args.cpu_images
            c0 = time.perf_counter()
            for i in range(n):
                oracle_lossy(blobs[i % len(blobs)])
            cdt = time.perf_counter() - c0
            cpu = {"value": round(n * 2.0736 / cdt, 2), "unit": "MP/s", "cores": 1, "kind": "port",
                   "sample": f"{n} of the same 1080p files through oracle/jpeg_oracle.c (decode+IDCT+FDCT+quant+progressive optimal-Huffman), 1 thread, {cdt:.1f} s"}
            # the same port on every host core (the reference's rayon par_iter shape), and the libjpeg-turbo proxy (Pillow: decode to YCbCr,
            # re-encode q80 4:2:0 progressive + optimised tables -- the plain profile the oracle is pinned to) on every core
            from concurrent.futures import ThreadPoolExecutor
            cores = os.cpu_count() or 1
            m = min(max(cores * 8, n), 16 * n)
            c0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                list(ex.map(lambda i: len(oracle_lossy(blobs[i % len(blobs)])), range(m)))   # ctypes releases the GIL inside the oracle
            cdt = time.perf_counter() - c0
            cpu_all = {"value": round(m * 2.0736 / cdt, 2), "unit": "MP/s", "cores": cores, "kind": "port", "sample": f"{m} files, {cores} threads, {cdt:.1f} s"}
            c0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                list(ex.map(lambda i: len(pillow_proxy(blobs[i % len(blobs)])), range(m)))
            cdt = time.perf_counter() - c0
            cpu_pillow = {"value": round(m * 2.0736 / cdt, 2), "unit": "MP/s", "cores": cores, "kind": "libjpeg-turbo proxy (Pillow), not libcaesium",
                          "sample": f"{m} files, {cores} threads, {cdt:.1f} s"}
        other = None
        if extras:
            # the other single-GPU configurations of BASELINE.json, as sub-records (device time of the whole path, inputs resident in HBM):
            # configs[2] 3840x2160 RGB8 PNGs --lossless --png-opt-level 3; configs[3] the same 1080p JPEGs -> WebP q85 at long edge 1500
            other = {}
            try:
                from gen_synth import synth_png
                import multiprocessing as mp
                with mp.get_context("fork").Pool(4) as pool:
                    pngs = pool.starmap(synth_png, [(100 + k, 3840, 2160, "RGB") for k in range(4)])
                npng = 128   # k_png_huff is one wave per zlib stream (a latency of ~2.3 s for a 4K file, whatever the count): the batch must be wide
                pp = pkg.default_parameters(png_optimize=True, png_optimization_level=3)
                warm = api.png_batch(pngs[:2], pp, device=local); warm.run(); warm.close()     # code objects and allocator warm; the timed batch is new
                pb = api.png_batch([pngs[k % 4] for k in range(npng)], pp, device=local)
                ptm = pb.run()                                                                  # its FIRST run: a second one would find the files already inflated
                pouts = pb.fetch()
                pn = api.png_kernel_names()
                pdom = max(range(len(pn)), key=lambda i: ptm.kernel_ms[i])
                other["configs[2] 4K PNG --lossless -o3"] = {
                    "files": npng, "value": round(npng * 3840 * 2160 / 1e6 / (ptm.total_ms / 1e3), 1), "unit": "MP/s", "device_ms": round(ptm.total_ms, 1),
                    "in_bytes": sum(len(pngs[k % 4]) for k in range(npng)), "out_bytes": sum(len(o) for o in pouts if isinstance(o, bytes)),
                    "dominant_kernel": pn[pdom], "dominant_ms": round(ptm.kernel_ms[pdom], 1),
                    "kernel_ms": {pn[i]: round(ptm.kernel_ms[i], 1) for i in range(len(pn)) if pn[i] and ptm.kernel_ms[i] >= 0.05},
                    "note": "the k_png_inflate slot (k_png_huff + k_png_lz77) is one wave per zlib stream: a latency, the same for 16 or 500 files; the other kernels scale with the file count"}
                pb.close()
            except Exception as e:   # a sub-record must not take the headline down
                other["configs[2] 4K PNG --lossless -o3"] = {"error": str(e)[:200]}
            try:
                nweb = 1024   # the macroblock and boolean-coder kernels are one wave per (picture, partition): 256 files leave the chip half empty
                wb = api.webp_batch([blobs[k % len(blobs)] for k in range(nweb)], pkg.default_parameters(webp_quality=85, width=1500), device=local)
                wb.run()
                wtm = wb.run()
                wdom = max(range(len(names)), key=lambda i: wtm.kernel_ms[i])
                other["configs[3] JPEG -> WebP q85 long edge 1500"] = {
                    "files": nweb, "value": round(nweb * 2.0736 / (wtm.total_ms / 1e3), 1), "unit": "source MP/s", "device_ms": round(wtm.total_ms, 1), "out_bytes": int(wtm.out_bytes),
                    "dominant_slot": "WebP tail (Lanczos, RGB -> YUV, k_webp_mb, k_webp_stats, k_webp_code: one timing slot; split in profiles/r02_webp_kernel_stats_batch1024.csv)",
                    "dominant_ms": round(wtm.kernel_ms[wdom], 1)}
                wb.close()
            except Exception as e:
                other["configs[3] JPEG -> WebP q85 long edge 1500"] = {"error": str(e)[:200]}
        if extras and args.boundary_files > 0:
            # the boundary itself: cs_batch_compress, host buffers in -> host buffers out (marker parse, pinned upload, kernels, download);
            # PCIe and the host side are inside this number and never inside `value`
            nb = min(args.boundary_files, len(blobs))
            api.cs_batch_compress(blobs[:min(nb, 64)], params, device=local)     # warm the block caches as a long-running caller would have
            tm = []
            res = api.cs_batch_compress(blobs[:nb], params, device=local, timing=tm)
            ok = sum(1 for r in res if isinstance(r, bytes))
            boundary = {"entry": "cs_batch_compress", "files": nb, "ok": ok, "seconds": round(tm[0], 4), "files_per_s": round(nb / tm[0], 1),
                        "value": round(nb * 2.0736 / tm[0], 1), "unit": "MP/s", "note": "host buffers in and out, one call, second call of the process"}
        out = {
            "metric": "megapixels/sec JPEG q=80 1920x1080 batch", "value": round(value, 1), "unit": "MP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 1920x1080 q92 4:2:0 baseline JPEGs -> -q 80 progressive, inputs resident in HBM",
                       "files_per_gpu_per_step": args.batch, "unique_images": args.unique, "sharding": f"files/{world} ranks, no collective",
                       "profile": profile},
            "parity_spot_check": bool(parity),
            "device_ms_per_step": round(sum(tm.total_ms for tm in timings) / len(timings), 3),
            "kernel_ms": {names[i]: round(kms[i], 4) for i in range(len(names)) if names[i]},
            "roofline": roof, "phases": phases, "plain_profile": plain, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all, "cpu_proxy_pillow": cpu_pillow, "boundary": boundary, "other_configs": other,
            "host": {"nproc": os.cpu_count(), "cpu": cpu_model()},
            "bytes": {"in": int(t.in_bytes), "out": int(t.out_bytes), "coef_one_way": int(t.coef_bytes)},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
