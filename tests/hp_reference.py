"""Extended-precision (x87 80-bit `numpy.longdouble`) transient of SMALL circuits, same algorithm as the reference
(simulateTRAN.ts:146-238) but with ~3.5 more digits: an arbiter for the cases where the fp64
reference and the GPU path differ by more than the parity budget (which of the two carries the rounding error?).
Test infrastructure; pure Python loops, only for circuits with a few dozen unknowns."""
import numpy as np

LD = np.longdouble
EPS = LD(1e-15)
VT = LD(0.02585)


def _solve(A, b):
    n = len(b)
    A = [row[:] + [b[i]] for i, row in enumerate(A)]
    for k in range(n):
        imax = max(range(k, n), key=lambda i: (abs(A[i][k]), -i))
        if abs(A[imax][k]) < EPS:
            raise ZeroDivisionError("singular")
        A[k], A[imax] = A[imax], A[k]
        for i in range(k + 1, n):
            f = A[i][k] / A[k][k]
            if abs(f) < EPS:
                continue
            for j in range(k, n + 1):
                A[i][j] -= f * A[k][j]
    x = [LD(0)] * n
    for i in range(n - 1, -1, -1):
        s = A[i][n]
        for j in range(i + 1, n):
            s -= A[i][j] * x[j]
        x[i] = s / A[i][i]
    return x


def run(flat, steps, dt, src):
    """flat: abi.FlatCircuit (1 instance).  Returns out_v [steps+1][n_nodes] as float64 (rounded from long double)
    and the iteration counts."""
    nN, nV = flat.n_nodes, flat.nV
    n = nN + nV
    dt = LD(dt)
    dtc = max(dt, EPS)
    vprev = [LD(v) for v in flat.C_vprev[0]]
    iprev = [LD(v) for v in flat.L_iprev[0]]
    vdprev = [LD(v) for v in flat.D_vdprev[0]]
    ison = [bool(v) for v in flat.S_ison[0]]
    out = np.zeros((steps + 1, nN))
    iters = np.zeros(steps + 1, np.int32)

    def adm(A, n1, n2, Y):
        i1, i2 = n1 - 1, n2 - 1
        if i1 >= 0:
            A[i1][i1] += Y
        if i2 >= 0:
            A[i2][i2] += Y
        if i1 >= 0 and i2 >= 0:
            A[i1][i2] -= Y
            A[i2][i1] -= Y

    def cur(b, n1, n2, I):
        if n1 > 0:
            b[n1 - 1] -= I
        if n2 > 0:
            b[n2 - 1] += I

    def v(x, nd):
        return LD(0) if nd == 0 else x[nd - 1]

    x = [LD(0)] * n
    for step in range(steps + 1):
        it = 0
        for it in range(20):
            A = [[LD(0)] * n for _ in range(n)]
            b = [LD(0)] * n
            for i in range(flat.nR):
                adm(A, flat.R_n1[i], flat.R_n2[i], LD(1) / LD(flat.R_val[0, i]))
            for i in range(flat.nC):
                g = LD(flat.C_val[0, i]) / dtc
                adm(A, flat.C_n1[i], flat.C_n2[i], g)
                cur(b, flat.C_n1[i], flat.C_n2[i], -g * vprev[i])
            for i in range(flat.nL):
                g = dtc / LD(flat.L_val[0, i])
                adm(A, flat.L_n1[i], flat.L_n2[i], g)
                cur(b, flat.L_n1[i], flat.L_n2[i], iprev[i])
            for i in range(flat.nS):
                r = LD(flat.S_ron[0, i] if ison[i] else flat.S_roff[0, i])
                adm(A, flat.S_n1[i], flat.S_n2[i], LD(1) / max(abs(r), EPS))
            for k in range(nV):
                i1, i2, j = flat.V_n1[k] - 1, flat.V_n2[k] - 1, nN + k
                if i1 >= 0:
                    A[i1][j] += 1
                    A[j][i1] += 1
                if i2 >= 0:
                    A[i2][j] -= 1
                    A[j][i2] -= 1
                b[j] += LD(src[step, k])
            for i in range(flat.nD):
                vd = vdprev[i] if it == 0 else v(x, flat.D_np[i]) - v(x, flat.D_nm[i])
                vl = min(max(vd, LD(-1.0)), LD(0.8))
                nvt = LD(flat.D_n[0, i]) * VT
                e = np.exp(vl / nvt)
                Is = LD(flat.D_is[0, i])
                idd = Is * (e - 1)
                gd = max(Is / nvt * e, LD(1e-12))
                adm(A, flat.D_np[i], flat.D_nm[i], gd)
                cur(b, flat.D_np[i], flat.D_nm[i], idd - gd * vl)
            x = _solve(A, b)
            flipped = False
            for i in range(flat.nS):
                vc = v(x, flat.S_cp[i]) - v(x, flat.S_cn[i])
                if ison[i] and vc < LD(flat.S_voff[0, i]):
                    ison[i] = False
                    flipped = True
                elif not ison[i] and vc > LD(flat.S_von[0, i]):
                    ison[i] = True
                    flipped = True
            if not flipped:
                break
        iters[step] = it + 1
        out[step] = [float(x[i]) for i in range(nN)]
        for i in range(flat.nC):
            vprev[i] = v(x, flat.C_n1[i]) - v(x, flat.C_n2[i])
        for i in range(flat.nL):
            iprev[i] = dtc / LD(flat.L_val[0, i]) * (v(x, flat.L_n1[i]) - v(x, flat.L_n2[i])) + iprev[i]
        for i in range(flat.nD):
            vdprev[i] = v(x, flat.D_np[i]) - v(x, flat.D_nm[i])
    return out, iters


def run_ac(flat, freqs, vph):
    """Extended-precision dense complex solve of the AC system (simulateAC.ts:25-62 + partial pivoting) per frequency.
    Returns complex128 [n_freq][n_nodes] (rounded from 80-bit)."""
    CL = np.clongdouble
    J = CL(1j)
    nN, nV = flat.n_nodes, flat.nV
    n = nN + nV
    out = np.zeros((len(freqs), nN), np.complex128)
    for fi, f in enumerate(freqs):
        A = np.zeros((n, n), CL)
        b = np.zeros(n, CL)

        def adm(n1, n2, Y):
            i1, i2 = n1 - 1, n2 - 1
            if i1 >= 0:
                A[i1, i1] += Y
            if i2 >= 0:
                A[i2, i2] += Y
            if i1 >= 0 and i2 >= 0:
                A[i1, i2] -= Y
                A[i2, i1] -= Y

        w = LD(2 * 3.141592653589793) * LD(f)
        for i in range(flat.nR):
            adm(flat.R_n1[i], flat.R_n2[i], CL(LD(1) / LD(flat.R_val[0, i])))
        for i in range(flat.nC):
            adm(flat.C_n1[i], flat.C_n2[i], J * CL(w * LD(flat.C_val[0, i])))
        for i in range(flat.nL):
            adm(flat.L_n1[i], flat.L_n2[i], CL(1) / (J * CL(w * LD(flat.L_val[0, i]))))
        for k in range(nV):
            i1, i2, j = flat.V_n1[k] - 1, flat.V_n2[k] - 1, nN + k
            if i1 >= 0:
                A[i1, j] += 1
                A[j, i1] += 1
            if i2 >= 0:
                A[i2, j] -= 1
                A[j, i2] -= 1
            b[j] += CL(vph[k])
        M = np.concatenate([A, b[:, None]], axis=1)
        for k in range(n):
            im = k + int(np.argmax(np.abs(M[k:, k])))
            M[[k, im]] = M[[im, k]]
            for i in range(k + 1, n):
                if M[i, k] != 0:
                    M[i, k:] -= M[i, k] / M[k, k] * M[k, k:]
        x = np.zeros(n, CL)
        for i in range(n - 1, -1, -1):
            x[i] = (M[i, n] - M[i, i + 1:n] @ x[i + 1:]) / M[i, i]
        out[fi] = np.array(x[:nN], dtype=complex)
    return out
