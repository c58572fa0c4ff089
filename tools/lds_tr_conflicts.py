#!/usr/bin/env python3
"""Bank-conflict model of the weight gradient's transpose reads (wgrad_sp_kernel): 64 banks x 4 B, a 64-bit read of a wave is served
half-wave by half-wave (32 lanes x 8 B = 256 B per clock when conflict-free); two lanes conflict when they touch the same bank at
different dword addresses.  cost = sum over the two halves of the worst bank multiplicity (ideal: 2)."""
import itertools

def cost(addrs):   # addrs: 64 byte addresses (8-byte reads)
    tot = 0
    for h in range(2):
        banks = {}
        for a in addrs[32 * h:32 * h + 32]:
            for d in (a // 4, a // 4 + 1):
                banks.setdefault(d % 64, set()).add(d)
        tot += max(len(v) for v in banks.values())
    return tot

def slot(c, P): return (c & 3) * P + (c >> 2)

def reads(TH, TW, CB, P_dy, P_in, plane_pad_dy, plane_pad_in, rot=2):
    """all distinct (image, K-step-independent) read patterns of one wave: dy rd 0/1, input rd 0/1 x kx 0..2; returns list of costs"""
    out = {}
    def img(ROWS, P, pad):
        RS = 4 * P + rot; ROWB = RS * 16
        PLANE = ((ROWS * ROWB + 255) // 256) * 256 + pad
        return ROWB, PLANE
    ROWB_d, PL_d = img(TH, P_dy, plane_pad_dy)
    ROWB_i, PL_i = img(TH + 2, P_in, plane_pad_in)
    for cot in range(2 if CB == 32 else 1):
        for rd in range(2):
            a = []
            for lane in range(64):
                sup, grp = lane & 15, lane >> 4
                pk = 8 * grp + 4 * rd + (sup >> 2)
                row, col, o, bo = pk // TW, pk % TW, (sup & 3) >> 1, (sup & 1) * 8
                a.append((2 * cot + o) * PL_d + row * ROWB_d + slot(col, P_dy) * 16 + bo)
            out[("dy", cot, rd)] = cost(a)
            for kx in range(3):
                a = []
                for lane in range(64):
                    sup, grp = lane & 15, lane >> 4
                    pk = 8 * grp + 4 * rd + (sup >> 2)
                    row, col, o, bo = pk // TW, pk % TW, (sup & 3) >> 1, (sup & 1) * 8
                    a.append((2 * cot + o) * PL_i + row * ROWB_i + slot(col + kx + 3, P_in) * 16 + bo)
                out[("in", cot, rd, kx)] = cost(a)
    return out

if __name__ == "__main__":      # CPU only: python tools/lds_tr_conflicts.py   (profiles/r3_wgrad_sp_lds_conflicts.md)
    for TH, TW in ((4, 32), (8, 16)):
        cur = reads(TH, TW, 32, TW // 4, (TW + 8) // 4, 0, 0)
        print(f"tile {TH}x{TW}: current layout (P = NQ, PLANE = 0 mod 256): mean {sum(cur.values()) / len(cur):.2f} clocks per "
              f"transpose read of a wave (ideal 2.00); worst {max(cur.values())}")
        best = []
        for Pd, Pi, pd, pi in itertools.product((4, 8, 12, 20), (6, 10, 12, 20), range(0, 256, 16), range(0, 256, 16)):
            if Pd < TW // 4 or Pi < (TW + 8) // 4:
                continue
            r = reads(TH, TW, 32, Pd, Pi, pd, pi)
            best.append((sum(r.values()) / len(r), max(r.values()), Pd, Pi, pd, pi))
        best.sort()
        print("  best (mean, worst, P dy, P in, plane pad dy, plane pad in):", best[:3])
