"""Per-file sharding across GPUs: what caesium-clt's rayon `par_iter` over files
(/root/reference/src/compressor.rs:81-100) becomes with one process per GPU.

Files are independent, so the partition is static round-robin (file i -> rank i % world), there is no collective on the
data path, and the only exchange is the final gather of results, which restores INPUT ORDER (the reference's
order-preserving `collect()`, asserted by its tests at compressor.rs:789-792)."""


def shard_indices(n_files, rank, world):
    return list(range(rank, n_files, world))


def compress_sharded(api, blobs, params, rank, world, device=0, group=None, fmt=None):
    """every rank passes the same `blobs`; returns the full result list (input order) on every rank.  The shard goes through the C
    entry point itself (`cs_batch_compress`: JPEG and, under png_optimize, PNG files in one call) or, with `fmt`, through
    `cs_batch_convert`"""
    mine = shard_indices(len(blobs), rank, world)
    if not mine:
        outs = []
    elif fmt is None:
        outs = api.cs_batch_compress([blobs[i] for i in mine], params, device)
    else:
        outs = api.batch_convert([blobs[i] for i in mine], params, fmt, device)
    if world == 1:
        return outs
    import torch.distributed as dist
    gathered = [None] * world
    dist.all_gather_object(gathered, [(i, o if isinstance(o, bytes) else ("ERR", o.code, str(o))) for i, o in zip(mine, outs)], group=group)
    result = [None] * len(blobs)
    for part in gathered:
        for i, o in part:
            result[i] = o
    return result
