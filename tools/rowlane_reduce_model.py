#!/usr/bin/env python3
"""Symbolic model of the lane-reduce network of csrc/scan_bwdr.hip (rl_reduce_half + rl_reduce_finish): every lane
value is tracked as the set of (slot, source lane) terms it contains; the script checks that each of the 32 slots of a
tile (memory half hm, position j in the half, array dB / dC) ends up summed over all 64 lanes in exactly two lanes, and
prints the lane -> slot table the kernel uses (kSlotOfLane).  Lane-operation semantics as pinned on the device by
tools/ubench/lane_ops_probe.hip (permlane swaps) and by sigma_scan_selftest (the whole network).

    python tools/rowlane_reduce_model.py
"""
import numpy as np

NSLOT = 32


def unit(slot):
    m = np.zeros((64, NSLOT, 64), dtype=np.int64)           # [lane][slot][source lane]
    for lane in range(64):
        m[lane, slot, lane] = 1
    return m


def fold32(a, b):      # lanes 0-31 = a[0:32] + a[32:64], lanes 32-63 = b[0:32] + b[32:64]
    r = np.empty_like(a)
    r[:32] = a[:32] + a[32:]
    r[32:] = b[:32] + b[32:]
    return r


def fold16(a, b):      # DPP rows = {a.r0 + a.r1, b.r0 + b.r1, a.r2 + a.r3, b.r2 + b.r3}
    ar = [a[16 * i:16 * i + 16] for i in range(4)]
    br = [b[16 * i:16 * i + 16] for i in range(4)]
    return np.concatenate([ar[0] + ar[1], br[0] + br[1], ar[2] + ar[3], br[2] + br[3]])


def dpp_add(old, src, perm, bank_mask):     # old[lane] <- src[perm(lane)] + src[lane] where the bank is enabled and the source exists
    out = old.copy()
    for lane in range(64):
        row, l = divmod(lane, 16)
        s = perm(l)
        if s is None or not (0 <= s < 16) or not ((bank_mask >> (l // 4)) & 1):
            continue
        out[lane] = src[16 * row + s] + src[lane]
    return out


ror8 = lambda l: (l + 8) % 16
shl4 = lambda l: l + 4
shr4 = lambda l: l - 4


def quad(a, perm):
    r = np.empty_like(a)
    for lane in range(64):
        r[lane] = a[(lane & ~3) + perm[lane & 3]]
    return r


def slot_id(hm, j, arr):
    return (hm * 8 + j) * 2 + arr


def reduce_half(hm):
    """16 terms of one half -> one register (each lane: the sum over 16 lanes of one slot)"""
    w = [fold32(unit(slot_id(hm, j, 0)), unit(slot_id(hm, j, 1))) for j in range(8)]
    z = [fold16(w[0], w[4]), fold16(w[1], w[5]), fold16(w[2], w[6]), fold16(w[3], w[7])]
    zero = np.zeros_like(z[0])
    q0 = dpp_add(dpp_add(zero, z[0], ror8, 0x3), z[1], ror8, 0xC)
    q1 = dpp_add(dpp_add(zero, z[2], ror8, 0x3), z[3], ror8, 0xC)
    return dpp_add(dpp_add(zero, q0, shl4, 0x5), q1, shr4, 0xA)


def finish(p0, p1):
    t0 = p0 + quad(p0, [2, 3, 0, 1])
    t1 = p1 + quad(p1, [2, 3, 0, 1])
    o = np.where((np.arange(64) & 2)[:, None, None] != 0, t1, t0)
    return o + quad(o, [1, 0, 3, 2])


def main():
    f = finish(reduce_half(0), reduce_half(1))
    table = []
    for lane in range(64):
        slots = np.nonzero(f[lane].sum(axis=1))[0]
        assert len(slots) == 1, (lane, slots)
        assert (f[lane, slots[0]] == 1).all(), lane
        table.append(int(slots[0]))
    assert sorted(table) == sorted(list(range(32)) * 2)
    print("slot = (hm * 8 + j) * 2 + arr; lane -> slot:")
    print(table)
    # memory offset inside the tile and array of each lane
    print("lane -> (arr, memory position):", [(s & 1, s >> 1) for s in table])


if __name__ == "__main__":
    main()
