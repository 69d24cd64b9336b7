"""LDS bank-conflict model for the fragment reads of the window / pair / ResBlock / attention kernels (no GPU needed).

MI355X guide, LDS section: a wave64 `ds_read_b128` is served in four groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
and the same + 32 — one LDS cycle per group when the group's 16 x 16 bytes hit 64 distinct banks (bank = (address / 4) mod 64); every
extra distinct address on a bank adds a cycle.  A B / A fragment read of the 16 x 16 x 32 MFMA: lane (q = lane >> 4, l15 = lane & 15)
reads 16 bytes at row R0 + l15, 16-byte chunk 4 ks + q (optionally swizzled).  `cycles` = LDS cycles of one such wave instruction
(4 = conflict-free), worst case over the row offset R0 (every tap shift).

Round 4 used it to find that the C * 2 + 16 row stride of conv_pair_fs_kernel made every fragment read two-way conflicted (48 % of
the kernel's LDS cycles), that strides of 2 mod 4 sixteen-byte units are conflict-free at every shift, that resblock.hip's XOR
swizzles are clean, and that flash_attn.hip's `store_tr8` image also serves row-major fragment reads without conflicts."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def cycles(addr_of_lane, width=16):
    tot = 0
    for g in GROUPS:
        banks = {}
        for l in g:
            a = addr_of_lane(l)
            for b in range(width // 4):
                banks.setdefault(((a // 4) + b) % 64, set()).add(a)
        tot += max(len(v) for v in banks.values())
    return tot


def frag(row_stride, swizzle=None, ks=0, shifts=range(64)):
    worst = 0
    for R0 in shifts:
        def addr(l):
            q, l15 = l >> 4, l & 15
            row, ch = R0 + l15, ks * 4 + q
            if swizzle:
                return swizzle(row, ch)
            return row * row_stride + ch * 16
        worst = max(worst, cycles(addr))
    return worst


if __name__ == "__main__":
    print("pair kernel, C = 64: row stride 144 B (round 3):", frag(144), "| 160 B (round 4):", frag(160), frag(160, ks=1))
    print("pair kernel, C = 32: row stride  80 B (round 3):", frag(80), "|  96 B (round 4):", frag(96))
    print("pair kernel, C = 128 (288 B):", frag(288), "| C = 256 / proj32 (544 B):", frag(544), "| win_conv CIN = 512 (1056 B):", frag(1056))
    print("resblock / mrf32, C = 32 (64-B rows, chunk ^ (row >> 1) & 3):", frag(64, lambda r, c: r * 64 + ((c ^ ((r >> 1) & 3)) << 4)))
    print("resblock, C = 64 (128-B rows, chunk ^ row & 7):", frag(128, lambda r, c: r * 128 + ((c ^ (r & 7)) << 4)),
          frag(128, lambda r, c: r * 128 + ((c ^ (r & 7)) << 4), ks=1))
    # flash_attn.hip store_tr8 image: [64 rows][256 B], 32-byte block b of row k at block b ^ (k & 7); a row-major fragment read of it
    # (rows R0 + l15 with R0 a multiple of 16, chunk c = 4 ks + q -> block c >> 1, half c & 1)
    tr8 = lambda r, c: r * 256 + ((((c >> 1) ^ (r & 7)) << 5) + ((c & 1) << 4))
    print("flash store_tr8 image read as rows (R0 multiples of 16):", [frag(0, tr8, ks=k, shifts=range(0, 64, 16)) for k in range(4)])
