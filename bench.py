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
        "k_pack": coef + t.out_bytes, "k_stats": coef, "k_sizes": coef,
        "k_xform_direct": 2 * y,
        "k_idct_plane": c + planes,
        "k_resample+k_plane_fdct": 3 * planes + c,
        "k_masks": coef + coef * 24 // 128,
        "memset_coef": 2 * coef,
    }
    return table.get(kernel)


# hipEvent kernel slot -> substring of the rocprofv3 kernel name (profiles/*.csv)
ROCPROF_NAME = {"k_dec_write": "k_dec_dense<2", "k_dec_spec": "k_dec_dense<0", "k_dec_relax0": "k_dec_dense<1", "k_dec_relax1_4": "k_dec_relax_list",
                "k_resample+k_plane_fdct": "k_resample_plane", "unstuff": "k_unstuff_copy", "k_emit": "k_emit_data"}


def pmc_traffic(kernel, batch):
    """HBM bytes of ONE launch of `kernel` from the committed PMC passes (profiles/r01_pmc_{FETCH,WRITE}_SIZE_batch<B>.csv:
    separate rocprofv3 --pmc runs of this same command at the same --batch, --steps 1; raw counter unit KiB; FETCH_SIZE doubled
    per the gfx950 note in MI355X_MICROARCH.md, WRITE_SIZE as is -- it reads exactly 2*coef bytes on the pool memset).
    None when the batch differs from the profiled one or the files are absent."""
    import csv
    key = ROCPROF_NAME.get(kernel, kernel)
    tot = 0.0
    for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        path = os.path.join(ROOT, "profiles", f"r01_pmc_{counter}_batch{batch}.csv")
        if not os.path.exists(path):
            return None
        hit = [r for r in csv.reader(open(path)) if len(r) == 3 and key in r[0]]
        if not hit:
            return None
        tot += sum(float(r[2]) / max(1, int(r[1])) for r in hit) * 1024.0 * scale
    return int(tot)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=2048, help="1080p files per rank per step (2048 x ~21 MB of device pools = 44 GB of the 288 GB)")
    ap.add_argument("--unique", type=int, default=16, help="distinct synthetic images (cycled to --batch)")
    ap.add_argument("--cpu-images", type=int, default=256, help="files timed through the CPU oracle (rank 0, N=1)")
    args = ap.parse_args()

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
    uniq = [synth_jpeg(rank * args.unique + i) for i in range(args.unique)]
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
    mp_per_step = t.pixels / 1e6 * world
    value = mp_per_step * args.steps / dt

    # spot-check parity on this very batch (outside the timed region)
    outs = batch.fetch()
    from _util import oracle_lossy
    parity = all(outs[i] == oracle_lossy(blobs[i]) for i in range(min(2, args.unique)))

    out = None
    if rank == 0:
        names = api.kernel_names()
        kms = [sum(tm.kernel_ms[i] for tm in timings) / len(timings) for i in range(len(names))]
        dom = max(range(len(names)), key=lambda i: kms[i])
        ab = algorithmic_bytes(names[dom], t, args.batch)
        roof = {"bound": "hbm", "kernel": names[dom], "avg_ms": round(kms[dom], 4), "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": pmc_traffic(names[dom], args.batch)}
        if ab is not None:
            ach = ab / (kms[dom] * 1e-3) / 1e9
            roof.update({"achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(ab)})
        else:
            roof.update({"achieved": None, "frac": None})
        cpu = None
        if world == 1 and args.cpu_images > 0:
            n = args.cpu_images
            c0 = time.perf_counter()
            for i in range(n):
                oracle_lossy(blobs[i % len(blobs)])
            cdt = time.perf_counter() - c0
            cpu = {"value": round(n * 2.0736 / cdt, 2), "unit": "MP/s", "cores": 1, "kind": "port",
                   "sample": f"{n} of the same 1080p files through oracle/jpeg_oracle.c (decode+IDCT+FDCT+quant+progressive optimal-Huffman), 1 thread, {cdt:.1f} s"}
        out = {
            "metric": "megapixels/sec JPEG q=80 1920x1080 batch", "value": round(value, 1), "unit": "MP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 1920x1080 q92 4:2:0 baseline JPEGs -> -q 80 progressive, inputs resident in HBM",
                       "files_per_gpu_per_step": args.batch, "unique_images": args.unique, "sharding": f"files/{world} ranks, no collective"},
            "parity_spot_check": bool(parity),
            "device_ms_per_step": round(sum(tm.total_ms for tm in timings) / len(timings), 3),
            "kernel_ms": {names[i]: round(kms[i], 4) for i in range(len(names)) if names[i]},
            "roofline": roof, "cpu_baseline": cpu,
            "bytes": {"in": int(t.in_bytes), "out": int(t.out_bytes), "coef_one_way": int(t.coef_bytes)},
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
