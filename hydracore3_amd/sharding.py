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
