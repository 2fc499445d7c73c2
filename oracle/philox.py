"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
as 1, 2, 3", SC'11; the generator of Random123 / cuRAND / torch's CUDA RNG) and of the loss mask the step prologue draws with it
(countr_step_prologue, include/countr_hip.h; replaces the per-iteration np.random.binomial(1, 0.8, [384, 384]) of the reference,
FSC_finetune_cross.py:290-292 -- same distribution, a counter-based stream instead of numpy's global Mersenne twister).
Pinned by the known-answer vectors of the Random123 distribution (tests/test_host_cpu.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: (k0, k1) -> uint32 [..., 4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & MASK32, (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & MASK32]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=-1).astype(np.uint32)


def loss_mask(seed, t, n=384 * 384, p_keep=0.8):
    """Mask number t (0-based) of a step object with mask_seed = seed: element 4 g + j = word j of Philox(counter (g, 0, t_lo, t_hi),
    key (seed_lo, seed_hi)) < floor(p_keep 2^32) -> float32 [n]."""
    g = np.arange(n // 4, dtype=np.uint32)
    ctr = np.stack([g, np.zeros_like(g), np.full_like(g, t & 0xFFFFFFFF), np.full_like(g, (t >> 32) & 0xFFFFFFFF)], axis=-1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    thr = min(int(p_keep * 4294967296.0), 0xFFFFFFFF)
    return (r.reshape(-1) < np.uint32(thr)).astype(np.float32)
