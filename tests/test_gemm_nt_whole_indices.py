"""Replay, on the CPU, of the index arithmetic of the whole-output weight-gradient kernels (csrc/ctn_gemm_nt_whole.hip):
for the shapes of the GPU tests and of the benchmark, every byte a lane may read lies inside its tensor and its row
pitch, every LDS address inside its plane buffer, every workspace index inside the size ctn_gemm_nt_ws_floats reports,
and no two staging threads store to the same bytes.  The lanes the kernel masks (pieces that start at or behind frame
T, stages behind the last one) get an offset outside the buffer descriptor's range and are not modelled as reads."""
import numpy as np
import pytest

def audit(BWD, FR, WM, WN, NWM, NWN, M, N, T, batch, M1, lda, ldab, lda2, ldb, rowsA, rowsAb, rowsB, abs_, abbs, a2bs, bbs, cus=256, late=None):
    WNT = 512
    MB, NB = 32*WM*NWM, 32*WN*NWN
    RB, CH = 2*FR, FR//8
    SWS = 3 if CH == 2 else 2
    KS, PPR = FR//16, FR//4
    RPP = WNT//PPR
    APLANE, BPLANE = MB*RB, NB*RB
    ABUF, BBUF = 3*APLANE, 3*BPLANE
    AQ, BQ = MB//RPP, NB//RPP
    tiles_m, tiles_n = M//MB, N//NB
    tiles = tiles_m*tiles_n
    spb = (T+FR-1)//FR
    nstages = batch*spb
    nz = max(1, cus//tiles); nz = min(nz, (nstages+3)//4, 65535)
    PS = MB*NB + MB
    need = nz*tiles*PS
    shm = 2*3*2*FR*(MB+NB)
    assert shm + 64 <= 163840, shm
    tid = np.arange(WNT); srow, sfq = tid//PPR, tid % PPR
    lane = tid & 63; l31 = lane & 31; half = lane >> 5; wave = tid >> 6
    sswz = (srow >> SWS) & (CH-1)
    st_off = srow*RB + (((sfq >> 1) ^ sswz)*16) + (sfq & 1)*8
    # LDS stores
    for q in range(AQ):
        for pl in range(3):
            o = st_off + q*RPP*RB + pl*APLANE
            assert o.min() >= 0 and o.max()+8 <= ABUF, ("A lds", o.max())
    for q in range(BQ):
        for pl in range(3):
            o = st_off + q*RPP*RB + pl*BPLANE
            assert o.max()+8 <= BBUF, ("B lds", o.max())
    # distinct store addresses (no two threads write the same 8 bytes of a pass)
    assert len(np.unique(st_off)) == WNT
    lswz = (l31 >> SWS) & (CH-1)
    wm, wn = wave % NWM, wave // NWM
    for ks in range(KS):
        coff = ((2*ks+half) ^ lswz)*16
        for i in range(WM):
            o = (32*WM*wm + l31)*RB + coff + 2*APLANE + i*32*RB
            assert o.max()+16 <= ABUF
        for j in range(WN):
            o = (32*WN*wn + l31)*RB + coff + 2*BPLANE + j*32*RB
            assert o.max()+16 <= BBUF
    # fragment <-> store consistency: the 16 bytes a lane reads for (row r, k-step ks, half) hold frames 8*(2ks+half)..+7 of row r
    # store: piece p (frames 4p..4p+3) of row r at r*RB + ((p>>1)^swz(r))*16 + (p&1)*8 ; read chunk c=2ks+half at r*RB + (c^swz(r))*16 -> pieces 2c, 2c+1 ok
    # global loads
    voA = lambda ld: (srow*ld + 4*sfq)*4
    worst = 0
    for tm in range(tiles_m):
        m0 = tm*MB
        for tn in range(tiles_n):
            n0 = tn*NB
            for bz in (0, 1, nz//2, nz-1):
                nb, ns = bz//spb, bz % spb
                nst = (nstages - bz + nz - 1)//nz
                for g in range(nst+3):
                    live = nb < batch
                    b = nb if live else 0
                    t0 = ns*FR if live else T
                    so = t0*4 if live else 0
                    ns += nz
                    while ns >= spb: ns -= spb; nb += 1
                    inn = t0 + 4*sfq < T
                    for q in range(AQ):
                        m = m0 + RPP*q
                        low = BWD or m < M1
                        if low:
                            off = voA(lda) + m*lda*4 + so
                            size = rowsA*lda*4
                            base = b*abs_*4
                            tot = batch*abs_*4 if abs_ else size
                        else:
                            off = voA(ldab) + (m-M1)*ldab*4 + so
                            size = rowsAb*ldab*4
                            base = b*abbs*4
                        o = off[inn]
                        if len(o):
                            assert o.min() >= 0 and o.max()+16 <= size, ("A", tm, tn, bz, g, q, o.max()+16, size)
                            # the row's own extent: frames < ld
                        if BWD:
                            off2 = voA(lda2) + m*lda2*4 + so
                            o = off2[inn]
                            if len(o): assert o.max()+16 <= M*lda2*4
                    for q in range(BQ):
                        off = voA(ldb) + (n0+RPP*q)*ldb*4 + so
                        o = off[inn]
                        if len(o): assert o.min() >= 0 and o.max()+16 <= rowsB*ldb*4, ("B", o.max()+16, rowsB*ldb*4)
                    # frames read stay inside the row pitch
                    fr = (t0 + 4*sfq)[inn]
                    if len(fr): assert fr.max()+4 <= min(lda, ldb, lda2 if BWD else lda, ldab if (not BWD and M1 < M) else lda), ("pitch", fr.max()+4)
    # workspace
    for bz in (0, nz-1):
        for t in range(tiles):
            base = (bz*tiles + t)*PS
            idx = (((7*WM + WM-1)*WN + WN-1)*16 + 15)*64 + 63
            assert idx < MB*NB and base + idx < need
            assert base + MB*NB + (RPP*(AQ-1) + srow.max()) < need
    # reduction kernel
    NBLK = MB*NB//256
    zper = (nz+3)//4
    for blk in (0, NBLK-1):
        p = blk*256 + 63*4 + 3
        assert p < MB*NB
        assert (tiles-1)*PS + (nz-1)*tiles*PS + p < need
    for blk in range(MB//64):
        r = blk*64 + 63
        assert r < MB and (tiles-1)*PS + (nz-1)*tiles*PS + MB*NB + r < need
    return dict(nz=nz, need=need, shm=shm, spb=spb)


def r4(n):
    return (n + 3) // 4 * 4


@pytest.mark.parametrize("M1,M2,N,T,B", [(128, 128, 512, 3999, 5), (128, 128, 256, 8191, 4), (256, 0, 512, 4100, 4),
                                          (128, 128, 512, 4001, 5), (128, 128, 512, 3999, 6), (128, 128, 512, 3999, 16),
                                          (128, 128, 512, 3999, 32)])
def test_heads_form_indices(M1, M2, N, T, B):
    ld = r4(T)
    r = audit(False, 16, 4, 2, 2, 4, M1 + M2, N, T, B, M1, ld, ld, ld, ld, M1, M2, N, M1 * ld, M2 * ld, 0, N * ld)
    assert r["need"] == 16842752 and r["shm"] == 98304


def test_heads_form_indices_on_strided_views():
    audit(False, 16, 4, 2, 2, 4, 256, 512, 3000, 6, 128, 3008, 3008, 3008, 3008, 128, 128, 512, 264 * 3008, 264 * 3008,
          0, 516 * 3008)


@pytest.mark.parametrize("H,NB,T,B", [(512, 128, 3999, 5), (256, 128, 5000, 7), (512, 128, 4001, 5), (512, 128, 3999, 16),
                                       (512, 128, 3999, 32)])
def test_bottleneck_form_indices(H, NB, T, B):
    ld = r4(T)
    r = audit(True, 32, 2, 2, 4, 2, H, NB, T, B, H, ld, ld, ld, ld, H, 0, NB, H * ld, 0, H * ld, NB * ld)
    assert r["need"] == 8454144 and r["shm"] == 147456


@pytest.mark.parametrize("M,N,T,B", [(1024, 128, 3999, 6), (1024, 128, 3999, 32), (512, 128, 4100, 9)])
def test_tall_form_indices(M, N, T, B):
    ld = r4(T)
    audit(False, 32, 2, 2, 4, 2, M, N, T, B, M, ld, ld, ld, ld, M, 0, N, M * ld, 0, 0, N * ld)
