"""Replay, on the CPU, of the index arithmetic of the row products' whole-output weight-gradient kernel
(rows_wgrad_b3_full_kernel in csrc/ctn_rows.hip): a lane's load is performed only when it lies inside the stage's buffer
descriptor, and every descriptor ends with the last valid row of its operand; the staging stores of a stage land on
distinct bytes inside their plane, and the transposing fragment reads stay inside the plane."""
import numpy as np
import pytest


def img(row, col, pitch):
    return row * pitch + (col >> 7) * 256 + 16 * (((col & 127) >> 3) ^ (((row & 3) << 2) | ((row >> 2) & 3))) + (col & 4) * 2


def audit(WM, WK, NWM, NWK, R, M, K, lddy, ldx, cus=256):
    NTHF, RS = 512, 16
    MB, KB = 32 * WM * NWM, 32 * WK * NWK
    YPITCH, XPITCH = 2 * MB, (2 * KB if KB >= 128 else 256)
    YPLANE, XPLANE = RS * YPITCH, RS * XPITCH
    YQ, XTH = RS * (MB // 4) // NTHF, RS * (KB // 4)
    assert YQ >= 1 and XTH <= NTHF and M % MB == 0 and K % KB == 0
    tid = np.arange(NTHF)
    YF4, XF4 = MB // 4, KB // 4
    yrow, ycol = tid // YF4, (tid % YF4) * 4
    xrow, xcol = tid // XF4, (tid % XF4) * 4
    YRQ = NTHF // YF4
    voY, voX, qY = (yrow * lddy + ycol) * 4, (xrow * ldx + xcol) * 4, YRQ * lddy * 4
    tiles = (M // MB) * (K // KB)
    nstages = (R + RS - 1) // RS
    nz = max(1, min(cus // tiles, nstages))
    ybytes, xbytes = R * lddy * 4, R * ldx * 4            # operand extents from their first element
    for m0 in range(0, M, MB):
        for k0 in range(0, K, KB):
            for bz in (0, nz - 1):
                nst = (nstages - bz + nz - 1) // nz
                for g in (0, 1, nst - 2, nst - 1, nst, nst + 1, nst + 2):
                    if g < 0:
                        continue
                    row0 = (g * nz + bz) * RS
                    left = R - row0
                    rows = RS if left >= RS else max(left, 0)
                    r0c = row0 if rows > 0 else 0
                    recY = ((rows - 1) * lddy + MB) * 4 if rows > 0 else 0
                    recX = ((rows - 1) * ldx + KB) * 4 if rows > 0 else 0
                    baseY, baseX = (r0c * lddy + m0) * 4, (r0c * ldx + k0) * 4
                    for q in range(YQ):
                        off = voY + q * qY
                        done = off + 16 <= recY                                    # the hardware's range check
                        assert (baseY + off[done] + 16 <= ybytes).all() and (off[done] >= 0).all()
                        # every element of a valid row is fetched
                        assert done[(yrow + YRQ * q) < rows].all() and not done[(yrow + YRQ * q) >= rows].any()
                    offx = voX[:XTH]
                    done = offx + 16 <= recX
                    assert (baseX + offx[done] + 16 <= xbytes).all()
                    assert done[xrow[:XTH] < rows].all() and not done[xrow[:XTH] >= rows].any()
    # LDS images
    seen = set()
    for q in range(YQ):
        o = img(yrow + YRQ * q, ycol, YPITCH)
        assert o.min() >= 0 and o.max() + 8 <= YPLANE
        seen.update(o.tolist())
    assert len(seen) == YQ * NTHF
    ox = img(xrow[:XTH], xcol[:XTH], XPITCH)
    assert ox.max() + 8 <= XPLANE and len(set(ox.tolist())) == XTH
    lane = np.arange(64)
    gq, gp, gblk, ghalf = (lane & 15) >> 2, lane & 3, ((lane >> 4) & 1) * 16, lane >> 5
    for c0, pitch, plane in [(32 * t, YPITCH, YPLANE) for t in range(MB // 32)] + [(32 * t, XPITCH, XPLANE) for t in range(KB // 32)]:
        for x in range(2):
            o = img(8 * ghalf + 4 * x + gq, c0 + gblk + 4 * gp, pitch)
            assert o.max() + 8 <= plane
    return nz


@pytest.mark.parametrize("R,lddy,ldx", [(511104, 512, 128), (5000, 512, 128), (4099, 520, 136), (16, 512, 128), (70000, 512, 132)])
def test_512_x_128(R, lddy, ldx):
    audit(4, 2, 4, 2, R, 512, 128, lddy, ldx)


@pytest.mark.parametrize("R,M,lddy,ldx", [(511104, 1024, 1024, 64), (4099, 1024, 1024, 64), (70000, 512, 512, 72), (16, 512, 512, 64)])
def test_512_x_64(R, M, lddy, ldx):
    audit(2, 2, 8, 1, R, M, 64, lddy, ldx)
