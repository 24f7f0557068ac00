#!/usr/bin/env python3
"""LDS bank-conflict check of the patch images of the conv / correlation kernels, and the small search
that found the Winograd images' placement.  CPU only.

Model (MI355X_MICROARCH.md, LDS): a wave64 access is served in fixed lane groups, one LDS cycle per group
when no two lanes of the group hit the same bank with different addresses.
    ds_read_b64    2 groups of 32 consecutive lanes, bank = (addr / 4) mod 64   (an 8-byte read = 2 banks)
    ds_read_b128   4 groups of 16: {0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32; bank as above (4 banks)
A group of a b64 / b128 read moves 256 bytes = every bank once, so "conflict-free" = the group's chunks are
pairwise different modulo 256 bytes.

    python tools/lds_swizzle_search.py            check the images the kernels use
    python tools/lds_swizzle_search.py --search   + the search over row offsets / half swaps for F(4x4,3x3)
"""
import itertools
import sys

B64_GROUPS = [list(range(0, 32)), list(range(32, 64))]
_G0 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
_G1 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
B128_GROUPS = [_G0, _G1, [l + 32 for l in _G0], [l + 32 for l in _G1]]


def worst(addr_of_lane, groups, width):
    """Largest number of distinct addresses of one group that share a bank column (1 = conflict-free)."""
    w = 0
    for g in groups:
        cols = {}
        for l in g:
            a = addr_of_lane(l)
            assert a % width == 0
            cols.setdefault((a % 256) // width, set()).add(a)
        w = max(w, max(len(v) for v in cols.values()))
    return w


# ---- F(4x4,3x3), wino43_kernel.h: lane l: tile t = l % 16 (ty = t >> 2, tx = t & 3), k-pair g = l / 16
#      (16-byte half g >> 1, 8-byte part g & 1); tap (r, c) of the 6x6 input tile of tile block tb:
#      pixel (4 ty + r, 16 tb + 4 tx + c) in 32-byte cell py * pitch + px + rowshift(py), halves swapped by swap(px)
def wino43_addr(l, r, c, tb, pitch=34, rowshift=lambda py: py >> 2, swap=lambda px: (px >> 3) & 1, tby=0):
    t, g = l % 16, l // 16
    py, px = 16 * tby + 4 * (t >> 2) + r, 16 * tb + 4 * (t & 3) + c
    cell = py * pitch + px + rowshift(py)
    return cell * 32 + (((g >> 1) ^ swap(px)) * 16) + (g & 1) * 8


def check_wino43(**kw):
    return max(worst(lambda l: wino43_addr(l, r, c, tb, **kw), B64_GROUPS, 8)
               for r in range(6) for c in range(6) for tb in range(2))


# ---- F(2x2,3x3), wino_kernels.h: tile t = l % 16 (ty = t >> 3, tx = t & 7), tap (r, c) of the 4x4 tile:
#      pixel (2 (2 wb + ty) + r, 2 tx + c), cell py * 19 + px + ((py >> 1) & 1), same half swap
def wino23_addr(l, r, c, wb):
    t, g = l % 16, l // 16
    py, px = 2 * (2 * wb + (t >> 3)) + r, 2 * (t & 7) + c
    cell = py * 19 + px + ((py >> 1) & 1)
    return cell * 32 + (((g >> 1) ^ ((px >> 3) & 1)) * 16) + (g & 1) * 8


# ---- transposed conv, deconv_kernel.h: pixel t = l % 16 of patch row 4 w + pr, column shift dj:
#      px = t + 1 - dj, cell py * 17 + px, same half swap
def deconv_addr(l, pr, dj, w):
    t, g = l % 16, l // 16
    py, px = 4 * w + pr, t + 1 - dj
    return (py * 17 + px) * 32 + (((g >> 1) ^ ((px >> 3) & 1)) * 16) + (g & 1) * 8


# ---- correlation.hip: ds_read_b128 of quad q of neighbourhood pixel (ly + 2 tr, lx + 2 j), pixel stride 80 B,
#      row pitch 24 pixels; lane -> (row, column) so that every b128 group is one row of 16 pixels
def corr_lane(tid, mapped):
    if not mapped:
        return tid // 16, tid % 16
    l5 = tid & 31
    a = l5 < 4 or 12 <= l5 < 16 or 20 <= l5 < 28
    lx = (l5 if l5 < 4 else l5 - 8 if l5 < 16 else l5 - 12) if a else (l5 - 4 if l5 < 12 else l5 - 8 if l5 < 20 else l5 - 16)
    return (tid >> 5) * 2 + (0 if a else 1), lx


def corr_addr(l, wave, tr, j, q, mapped):
    lr, lx = corr_lane(wave * 64 + l, mapped)
    ly = 4 * (lr >> 1) + (lr & 1)
    return ((ly + 2 * tr) * 24 + lx + 2 * j) * 80 + q * 16


def main():
    print('F(4x4,3x3) patch image (cell py*34 + px + (py>>2), halves swapped when (px>>3)&1): worst multiplicity %d'
          % check_wino43())
    print('  ... without the row offset: %d; without the half swap: %d'
          % (check_wino43(rowshift=lambda py: 0), check_wino43(swap=lambda px: 0)))
    print('  ... the same placement for a 32x32-pixel tile (tile-block rows 0 and 1): %d'
          % max(check_wino43(tby=0), check_wino43(tby=1)))
    w = max(worst(lambda l: wino23_addr(l, r, c, wb), B64_GROUPS, 8) for r in range(4) for c in range(4) for wb in range(4))
    print('F(2x2,3x3) patch image (cell py*19 + px + ((py>>1)&1), same swap): worst multiplicity %d' % w)
    w = max(worst(lambda l: deconv_addr(l, pr, dj, wv), B64_GROUPS, 8) for pr in range(5) for dj in range(2) for wv in range(4))
    print('transposed-conv patch image (cell py*17 + px, same swap): worst multiplicity %d' % w)
    for mapped in (False, True):
        w = max(worst(lambda l: corr_addr(l, wave, tr, j, q, mapped), B128_GROUPS, 16)
                for wave in range(4) for tr in range(6) for j in range(5) for q in range(4))
        print('correlation neighbourhood (80-byte pixels, 24-pixel rows), lanes %s: worst multiplicity %d'
              % ('mapped to the b128 groups' if mapped else 'in plain order', w))
    if '--search' in sys.argv:
        print('search: cell = py*pitch + px + (py >> k1) [& m], halves swapped by (px >> k2) & 1; conflict-free ones:')
        for pitch, k1, m, k2 in itertools.product((34, 35, 36), (0, 1, 2, 3, None), (1, 3, 255), (2, 3, 4, None)):
            rs = (lambda py: 0) if k1 is None else (lambda py, k1=k1, m=m: (py >> k1) & m)
            sw = (lambda px: 0) if k2 is None else (lambda px, k2=k2: (px >> k2) & 1)
            if check_wino43(pitch=pitch, rowshift=rs, swap=sw) == 1:
                print('  pitch %d, row offset (py >> %s) & %d, swap (px >> %s) & 1' % (pitch, k1, m, k2))


if __name__ == '__main__':
    main()
