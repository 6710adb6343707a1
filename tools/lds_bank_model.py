#!/usr/bin/env python3
"""The LDS bank model behind round 5's lane mappings ("LDS banks decide the lane mappings").

An LDS access of a wave is served 32 lanes at a time for 4-byte operands, 16 lanes at a time for 8-byte ones; a pass costs as many
cycles as its deepest bank is hit by DIFFERENT addresses (32 banks of 4 bytes; lanes reading one address are a broadcast).  The
function below is all there is to it.  The script prints, for the places where which lane takes which item is a free choice, the cost of
the candidate mappings -- the numbers quoted in DESIGN.md and profiles/r05_experiments.txt; the hardware's own count is
SQ_LDS_BANK_CONFLICT in profiles/*_pmc_summary.csv (front half 593 -> 273 conflict cycles per frame, synthesis 157 -> 30).

  python tools/lds_bank_model.py
"""
import collections


def pass_cost(addrs_dw, width_dw=1):
    """cycles of one LDS instruction: addrs_dw = per-lane start address in 4-byte words (None: lane inactive)"""
    lanes_per_pass = 32 // width_dw
    total = 0
    for s in range(0, len(addrs_dw), lanes_per_pass):
        banks, seen = collections.Counter(), set()
        for a in addrs_dw[s:s + lanes_per_pass]:
            if a is None or a in seen:
                continue
            seen.add(a)
            for w in range(width_dw):
                banks[(a + w) % 32] += 1
        if banks:
            total += max(banks.values())
    return total


def kissfft_factors(n):
    f, p = [], 4
    while n > 1:
        while n % p:
            p = 2 if p == 4 else (3 if p == 2 else p + 2)
            if p * p > n:
                p = n
        n //= p
        f.append((p, n))
    return f


def transform_middle_stages():
    print("transform, middle stages (loads + stores of 8-byte complex elements, twiddle reads): LDS cycles per stage")
    print("  old: neighbouring lanes on neighbouring butterflies of one sub-transform; new: on the same butterfly of neighbouring sub-transforms")
    for nfft in (240, 180, 160, 120, 90, 80, 60, 40, 30):
        fs = kissfft_factors(nfft)
        row = []
        for s, (p, m) in enumerate(fs):
            if s == 0 or s == len(fs) - 1:
                continue
            nb, nblk = nfft // p, nfft // (p * m)
            for name, mp in (("old", lambda u: (u // m, u % m)), ("new", lambda u: (u % nblk, u // nblk))):
                cost = ideal = 0
                for r0 in range(0, nb, 64):
                    lanes = list(range(r0, min(r0 + 64, nb)))
                    for k in range(p):
                        ad = [2 * (mp(u)[0] * p * m + mp(u)[1] + k * m) for u in lanes]
                        cost += 2 * pass_cost(ad, 2)
                        ideal += 2 * ((len(lanes) + 15) // 16)
                    fstr = nfft // (p * m)
                    for k in range(1, p):
                        cost += pass_cost([2 * ((mp(u)[1] * fstr * k) % nfft) for u in lanes], 2)
                        ideal += (len(lanes) + 15) // 16
                row.append("radix %d m %d %s %d (ideal %d)" % (p, m, name, cost, ideal))
        print("  nfft %3d  %s" % (nfft, "; ".join(row)))


def resampler():
    print("LTPF resampler at 48 kHz (p = 4): the first sample read of the 64 lanes, 4-byte reads; ideal 2")
    q = lambda n: (15 * n) // 4
    maps = {
        "old pairing, lane l -> outputs 2l, 2l + 1": [q(2 * l) for l in range(64)],
        "outputs n, n + 4; neighbouring lanes inside one block of 8": [q(8 * (l // 4) + l % 4) for l in range(64)],
        "outputs n, n + 4; neighbouring lanes on neighbouring blocks": [q(8 * (l % 16) + l // 16) for l in range(64)],
    }
    for name, base in maps.items():
        print("  %-62s %d" % (name, pass_cost(base)))


def tns_autocorrelation():
    print("TNS autocorrelation, full band at 48 kHz / 10 ms (sub-blocks start at 12, 61, 110 | 160, 240, 320): cycles per term, both operands; ideal 4")
    starts = [[12, 61, 110], [160, 240, 320]]

    def cost(assign):
        pa = [None if it is None else starts[it[0]][it[1]] for it in assign]
        pb = [None if it is None else starts[it[0]][it[1]] + it[2] for it in assign]
        return pass_cost(pa) + pass_cost(pb)

    old = [(l // 27, (l % 27) % 3, (l % 27) // 3) if l < 54 else None for l in range(64)]
    new = []
    for l in range(64):
        if (l & 31) >= 27:
            new.append(None)
            continue
        j, k = (l & 31) // 9, (l & 31) % 9
        blk = (3 if j == 2 else j) if l < 32 else (2 if j == 0 else j + 3)
        new.append((blk // 3, blk % 3, k))
    print("  lane = 27 f + 3 k + s (old): %d;  nine lags of a sub-block on neighbouring lanes, {f0s0, f0s1, f1s0 | f0s2, f1s1, f1s2} per half-wave: %d"
          % (cost(old), cost(new)))


if __name__ == "__main__":
    transform_middle_stages()
    resampler()
    tns_autocorrelation()
