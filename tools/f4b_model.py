#!/usr/bin/env python3
"""Index-level numpy model of csrc/wino4b_fused.hip (one workgroup, one item): the packed U image, the swizzled patch image and its
DMA slot decode, the lane -> (tile, channel pair) map of the input transform, the A / B operand gathers of the term-paired
v_mfma_f32_16x16x32_bf16, the accumulated output transform and the epilogue's (lane, register) -> pixel map -- every formula as the
kernel has it, so that an indexing change can be checked on the CPU before it goes to the GPU.
    python tools/f4b_model.py            # prints max |error| against a direct convolution (float64) for a few shapes"""
import numpy as np

G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)

ROWPITCH, CLS_BYTES, POS_BYTES, STAGE_BYTES = 2304, 39 * 1024, 6144, 18432


def bf16_rne(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    return (r << 16).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rne(x); r1 = (x - h).astype(np.float32)
    m = bf16_rne(r1); r2 = (r1 - m).astype(np.float32)
    return h, m, bf16_rne(r2)


def pack_u(w, nq, s):
    """stage images of (64-channel slice nq, 16-channel slice s): [k 12][j 3][term 3][n 64][c 16] as float32 values of the bf16 terms"""
    cin, cout = w.shape[2], w.shape[3]
    U = np.einsum("ia,abcn,jb->ijcn", G, w.astype(np.float64), G).astype(np.float32)     # [xi][nu][c][n]
    img = np.zeros((12, 3, 3, 64, 16), np.float32)
    for k in range(12):
        for j in range(3):
            xi, nu = k >> 1, 3 * (k & 1) + j
            t = split3(U[xi, nu, 16 * s:16 * s + 16, 64 * nq:64 * nq + 64].T)            # [n][c]
            for t3 in range(3):
                img[k, j, t3] = t[t3]
    return img


def patch_image(x, y0, x0, s):
    """LDS patch bytes as float32 words: two classes of 39 KiB; DMA slot decode as in patch_issue()"""
    H, W, _ = x.shape
    lds = np.zeros(2 * CLS_BYTES // 4, np.float32)
    for cls in range(2):
        for pc in range(39):
            for lane in range(64):
                slot = pc * 64 + lane
                row, rem = divmod(slot, 144)
                pxs, g = rem >> 2, rem & 3
                px = pxs ^ ((pxs >> 2) & 3)
                y, xx = y0 - 1 + 2 * row + cls, x0 - 1 + px
                ok = row < 17 and px < 34 and 0 <= y < H and 0 <= xx < W
                v = x[y, xx, 16 * s + 4 * g:16 * s + 4 * g + 4] if ok else np.zeros(4, np.float32)
                o = (cls * CLS_BYTES + pc * 1024 + lane * 16) // 4
                lds[o:o + 4] = v
    return lds


def produce(patch, k):
    """V stage image [j 3][term 3][tile 64][c 16] for stage k, by the kernel's lane map"""
    R, HF = k >> 1, k & 1
    V = np.zeros((3, 3, 64, 16), np.float32)
    for wave in range(8):
        for lane in range(64):
            ptx, pq = lane >> 3, lane & 7
            pl = wave * 2 * ROWPITCH + pq * 8
            t = np.zeros((6, 2), np.float32)
            for b in range(6):
                if b < HF or b > HF + 4:
                    continue
                px = 4 * ptx + b
                colb = (px ^ ((px >> 2) & 3)) * 64
                d = np.zeros((6, 2), np.float32)
                for a in range(6):
                    o = (pl + (a & 1) * CLS_BYTES + (a >> 1) * ROWPITCH + colb) // 4
                    d[a] = patch[o:o + 2]
                t[b] = (BT[R] @ d.astype(np.float64)).astype(np.float32)
            for j in range(3):
                v = (BT[3 * HF + j] @ t.astype(np.float64)).astype(np.float32)
                terms = split3(v)
                tile = 8 * wave + ptx
                for t3 in range(3):
                    V[j, t3, tile, 2 * pq:2 * pq + 2] = terms[t3]
    return V


def mfma_16x16x32(A, B):
    """A[lane 64][8], B[lane 64][8] -> D[lane][4]:  A[i = lane & 15][k = 8 (lane >> 4) + e],  B[k][j = lane & 15],  D[i = 4 (lane >> 4) + reg][j = lane & 15]"""
    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for lane in range(64):
        Am[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = A[lane]
        Bm[8 * (lane >> 4):8 * (lane >> 4) + 8, lane & 15] = B[lane]
    Dm = Am @ Bm
    D = np.zeros((64, 4))
    for lane in range(64):
        for r in range(4):
            D[lane, r] = Dm[4 * (lane >> 4) + r, lane & 15]
    return D


def run_item(x, w, bias, slope, y0, x0, nq):
    """outputs of the 32x32-pixel block at (y0, x0), channels 64 nq .. + 63: dict (y, x, ch) -> value"""
    cin = x.shape[2]
    Y = np.zeros((8, 64, 2, 4, 4, 4))            # [wave][lane][blk][reg i][a][j]
    for s in range(cin // 16):
        patch = patch_image(x, y0, x0, s)
        uimg = pack_u(w, nq, s)
        tmp = np.zeros((8, 64, 6, 2, 4))
        for k in range(12):
            V = produce(patch, k).reshape(3, 3, 64 * 32 // 2)        # per (j, term): bytes / 2 = bf16 elements, row-major [tile][16]
            Uk = uimg[k].reshape(3, 3, 64 * 16)
            for wave in range(8):
                wm, wn = wave & 3, wave >> 2
                lanes = np.arange(64)
                kg, r16 = lanes >> 4, lanes & 15
                lo = kg < 2
                a_row = ((16 * wm + r16) * 32 + (kg & 1) * 16) // 2     # in bf16 elements
                b_row = ((32 * wn + r16) * 32 + (kg & 1) * 16) // 2
                def gather(img, j, term, row):
                    return np.stack([img[j, term[l], row[l]:row[l] + 8] for l in range(64)])
                for j in range(3):
                    A12 = gather(V, j, np.where(lo, 0, 1), a_row); A21 = gather(V, j, np.where(lo, 1, 0), a_row); A13 = gather(V, j, np.where(lo, 0, 2), a_row)
                    for blk in range(2):
                        B12 = gather(Uk, j, np.where(lo, 0, 1), b_row + blk * 256); B31 = gather(Uk, j, np.where(lo, 2, 0), b_row + blk * 256)
                        acc = mfma_16x16x32(A13, B31) + mfma_16x16x32(A21, B12) + mfma_16x16x32(A12, B12)
                        tmp[wave, :, 3 * (k & 1) + j, blk] = acc
            if k & 1:
                R = k >> 1
                T = np.einsum("wlcbi,cj->wlbij", tmp, AT.T)           # T[j] = sum_c M'[c] A[c][j]
                Y += np.einsum("a,wlbij->wlbiaj", AT[:, R], T)
    out = {}
    for wave in range(8):
        wm, wn = wave & 3, wave >> 2
        for lane in range(64):
            kg, r16 = lane >> 4, lane & 15
            for blk in range(2):
                ch = nq * 64 + wn * 32 + blk * 16 + r16
                for i in range(4):
                    tl = 4 * kg + i
                    ty, tx = 2 * wm + (tl >> 3), tl & 7
                    for a in range(4):
                        for j in range(4):
                            z = Y[wave, lane, blk, i, a, j] + bias[ch]
                            out[(y0 + 4 * ty + a, x0 + 4 * tx + j, ch)] = max(z, z * slope)
    return out


def conv_ref(x, w, bias, slope):
    H, W, _ = x.shape
    xp = np.zeros((H + 2, W + 2, x.shape[2])); xp[1:-1, 1:-1] = x
    out = np.zeros((H, W, w.shape[3]))
    for dy in range(3):
        for dx in range(3):
            out += xp[dy:dy + H, dx:dx + W] @ w[dy, dx].astype(np.float64)
    out += bias
    return np.where(out > 0, out, out * slope)


if __name__ == "__main__":
    rs = np.random.RandomState(0)
    for (H, W, cin, cout, y0, x0, nq) in [(32, 32, 16, 64, 0, 0, 0), (40, 70, 32, 128, 32, 32, 1)]:
        x = rs.randn(H, W, cin).astype(np.float32)
        w = (rs.randn(3, 3, cin, cout) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
        b = rs.randn(cout).astype(np.float32)
        ref = conv_ref(x, w, b, 0.1)
        got = run_item(x, w, b, 0.1, y0, x0, nq)
        err = max(abs(v - ref[y, xx, ch]) for (y, xx, ch), v in got.items() if y < H and xx < W)
        n = sum(1 for (y, xx, ch) in got if y < H and xx < W)
        print("H %d W %d cin %d cout %d block (%d, %d) nq %d: %d outputs, max |err| %.3e" % (H, W, cin, cout, y0, x0, nq, n, err))
