"""How one frame is split over GPUs: contiguous windows of the tile-swizzled pixel index (`tid`) space.

Every (tid, pass) sample is independent and owns its RNG stream `m_randomGens[tid]` (integrator_pt.cpp:139,605), so a
rank that renders `tid in [begin, begin+count)` produces bit-identical pixels to the single-GPU run; windows are aligned
to one 8x8 tile (64 tids, integrator_rt.cpp:13-31) so that a wave's 64 lanes start on one tile. The frame is assembled
by summing the ranks' zero-initialised framebuffers (RCCL reduce over xGMI): the windows are disjoint, so the sum is exact.
"""


def tid_window(rank: int, world: int, n_threads: int, align: int = 64):
    """(tid_begin, tid_count) of `rank`; the windows of ranks 0..world-1 tile [0, n_threads) without overlap."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank / world size")
    per = ((n_threads + world - 1) // world + align - 1) // align * align
    begin = min(rank * per, n_threads)
    return begin, min(per, n_threads - begin)


CHUNK = 1024          # tids per interleaved chunk: 16 tiles of 8x8 pixels


def tid_interleave(rank: int, world: int, n_threads: int, chunk: int = CHUNK):
    """Interleaved split for load balance: rank r renders chunks r, r + world, r + 2*world, ... of `chunk` consecutive tids
    (path lengths vary systematically over the image -- sky / light / floor -- so contiguous bands finish at different times).
    Returns (tid_begin, item_count, chunk, stride) for hpt_set_tid_interleave + hpt_path_trace_block_dev; the union over ranks is
    [0, n_threads) without overlap."""
    if world <= 0 or not (0 <= rank < world) or chunk % 64 != 0:
        raise ValueError("bad rank / world size / chunk")
    n_chunks = (n_threads + chunk - 1) // chunk
    mine = range(rank, n_chunks, world)
    count = sum(min(chunk, n_threads - c * chunk) for c in mine)
    return rank * chunk, count, chunk, world


def interleaved_tids(rank: int, world: int, n_threads: int, chunk: int = CHUNK):
    """The (begin, count) runs of consecutive tids rank `rank` owns (what a renderer without the interleave mapping loops over)."""
    n_chunks = (n_threads + chunk - 1) // chunk
    return [(c * chunk, min(chunk, n_threads - c * chunk)) for c in range(rank, n_chunks, world)]
