"""N>1 host path: two processes (gloo, CPU) shard a file list round-robin, compress their shards (emulation build of the
kernels), gather, and must reproduce the single-process result in input order."""
import os
import subprocess
import sys
import textwrap

from _util import ROOT


WORKER = textwrap.dedent("""
    import os, sys, hashlib
    sys.path[:0] = [r'{root}', r'{root}/tools', r'{root}/tests']
    import torch.distributed as dist
    from _util import emul_api, package, oracle_lossy
    from gen_synth import synth_jpeg
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    pkg = package()
    from caesium_clt_amd.sharding import compress_sharded, shard_indices
    blobs = [synth_jpeg(i, 96 + 8 * i, 64 + 8 * (i % 3), texture=7 * i) for i in range(7)] + [b'garbage']
    assert shard_indices(8, rank, world) == list(range(rank, 8, world))
    res = compress_sharded(emul_api(), blobs, pkg.default_parameters(), rank, world)
    assert len(res) == 8 and res[7][0] == 'ERR' and res[7][1] == 10200
    for src, out in zip(blobs[:7], res[:7]):
        assert out == oracle_lossy(src)
    # a mixed --lossless shard (JPEG + PNG in one call per rank) and a conversion shard
    from _util import oracle_lossless, oracle_png, oracle_jpeg_to_webp
    from gen_synth import synth_png
    mixed = [blobs[0], synth_png(1, 40, 30, 'RGB', compress_level=1), blobs[1], synth_png(2, 33, 21, 'L', compress_level=1), synth_png(3, 20, 20, 'RGBA')]
    res = compress_sharded(emul_api(), mixed, pkg.default_parameters(jpeg_optimize=True, png_optimize=True, png_optimization_level=2), rank, world)
    want = [oracle_lossless(mixed[0]), oracle_png(mixed[1], 2), oracle_lossless(mixed[2]), oracle_png(mixed[3], 2), oracle_png(mixed[4], 2)]
    assert res == want
    res = compress_sharded(emul_api(), blobs[:5], pkg.default_parameters(webp_quality=70), rank, world, fmt=3)
    assert res == [oracle_jpeg_to_webp(b, 70) for b in blobs[:5]]
    # conversions whose sources are of both kinds (to WebP), and JPEG -> PNG
    from _util import oracle_jpeg_to_png
    from oracle import oracle as O
    res = compress_sharded(emul_api(), mixed[:4], pkg.default_parameters(webp_quality=60), rank, world, fmt=3)
    assert res == [oracle_jpeg_to_webp(mixed[0], 60), O.png_to_webp(mixed[1], 60), oracle_jpeg_to_webp(mixed[2], 60), O.png_to_webp(mixed[3], 60)]
    res = compress_sharded(emul_api(), blobs[:3], pkg.default_parameters(png_optimize=True, png_optimization_level=1), rank, world, fmt=1)
    assert res == [oracle_jpeg_to_png(b, True, 1) for b in blobs[:3]]
    dist.barrier()
    dist.destroy_process_group()
    print('rank', rank, 'ok')
""")


def test_two_rank_sharding_matches_oracle(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    from _util import emul_api
    emul_api()  # build once before the ranks race to do it
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o
