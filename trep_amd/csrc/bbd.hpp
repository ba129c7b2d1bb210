// bbd.hpp -- structured, pivot-search-free solve of the Newton / KKT system  [[Df11, -Dh1^T], [Dh2, 0]] x = f
// (reference: midpointvi.c:577-670 assembles it, :720-733 solves it with the dense pivoted LU of math-code.c:337-461).
//
// The reference's solver -- and gj_rows / gj_panel here -- treat the matrix as dense and search a pivot in every column.  For
// a tree mechanism neither is necessary: Df11 = -M/dt + O(dt) has the sparsity of the mass matrix (two configs couple only if
// a common body hangs below both), which for a branched tree is BORDERED BLOCK DIAGONAL: remove the configs of the trunk (the
// puppet's six torso configs) and the constraints, and what remains falls apart into independent blocks (the four limbs, four
// configs each) that couple only to the border.  And the Df11 part is mass-matrix dominated, i.e. safe to eliminate in any order
// without looking for a pivot.  The host finds that decomposition from the matrix's structural pattern (bbd_plan) and the
// system-specialised kernels solve along it (gj_bbd):
//
//   stage 1  every block ("group") on its own 16-lane DPP row, one matrix ROW per lane: the group's own rows plus the border
//            rows it touches, columns [own | border | rhs] in registers.  Gauss-Jordan over the own columns in index order; the
//            pivot row reaches the other lanes through `row_newbcast` DPP operands (the pivot lane is known at compile time:
//            no search, no v_readlane, no LDS).  All groups run in the same instructions.
//   stage 2  the border rows' accumulated Schur updates are added into the trailing (border x border) system through LDS
//            atomics; that dense system (puppet: 6 torso configs + 6 constraints) is eliminated the same way on one DPP row,
//            configs first, constraints last.
//   stage 3  the groups' own unknowns by back-substitution from the border solution.
//
// Serial chain: max own size + border size pivot steps (puppet: 4 + 12 = 16) instead of nf (28), each without the search.
// Every unpivoted pivot is guarded: |pivot| must exceed 2^-20 of its row's largest original entry (mass-dominated rows sit at
// O(1), the constraints' Schur pivots at O(dt^2 |Dh|^2 / m) relative to |Dh|).  If any guard fails -- or nothing was written yet
// -- the dense image is still untouched and the caller solves it with the pivoting solver (one trajectory = one wavefront, so
// that branch is uniform).  Solutions agree with the pivoting solvers' to rounding (the elimination order differs).
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

namespace tg {

// Packed plan tables (one copy per system; the kernels stage them in LDS once per launch):
//   tab[lane]            lane = 16 g + r:  (image row + 1) | (trailing index + 1) << 8 | (image index of trailing variable `lane` + 1) << 16
//                        rows r < ng are the group's own variables (0 = padding: an identity row), rows ng <= r < ng + nb its
//                        border variables (0 = none); the trailing index is that of a border row; the third field is set for lane < t
//   tab[64 + 16 g + j]   column j of group g:  (image column + 1) | (trailing index + 1) << 8   (j < ng own, ng <= j < ng + nb border)
struct BbdPlan {
    int ok = 0;
    int g = 0, ng = 0, nb = 0, t = 0;   // groups, largest own block, largest border list, trailing size
    int tvar[16];                       // image index of trailing variable i (configs first, then constraints)
    int tab[128];
    BbdPlan() { std::memset(tvar, 0, sizeof(tvar)); std::memset(tab, 0, sizeof(tab)); }
};

// pattern: nf x nf, non-zero where the matrix can be non-zero; variables 0 .. nd-1 are configs (unpivoted), nd .. nf-1 constraints.
inline BbdPlan bbd_plan(int nf, int nd, const std::vector<unsigned char> &pattern) {
    BbdPlan P;
    if (nf < 8 || nf > 255 || nd < 2) return P;
    auto adj = [&](int i, int j) { return pattern[(size_t)i * nf + j] || pattern[(size_t)j * nf + i]; };
    std::vector<char> border(nf, 0);
    for (int i = nd; i < nf; i++) border[i] = 1;
    std::vector<int> comp(nf, -1);
    int ncomp = 0;
    auto components = [&]() {
        std::fill(comp.begin(), comp.end(), -1);
        ncomp = 0;
        for (int s = 0; s < nd; s++) {
            if (border[s] || comp[s] >= 0) continue;
            std::vector<int> stack(1, s);
            comp[s] = ncomp;
            while (!stack.empty()) {
                const int u = stack.back(); stack.pop_back();
                for (int v = 0; v < nd; v++) if (!border[v] && comp[v] < 0 && adj(u, v)) { comp[v] = ncomp; stack.push_back(v); }
            }
            ncomp++;
        }
    };
    auto comp_rows = [&](int c, int &own, int &bord) {   // own variables and touched border variables of component c
        own = bord = 0;
        for (int i = 0; i < nd; i++) if (comp[i] == c) own++;
        for (int b = 0; b < nf; b++) {
            if (!border[b]) continue;
            bool touched = false;
            for (int i = 0; i < nd && !touched; i++) touched = comp[i] == c && adj(i, b);
            bord += touched ? 1 : 0;
        }
    };
    // grow the border until every component (with the border rows it touches) fits one 16-lane row: the config with the most
    // neighbours inside the largest offending component goes next (ties: the smaller index -- trunk configs come first)
    for (;;) {
        components();
        int worst = -1, worst_rows = 0;
        for (int c = 0; c < ncomp; c++) {
            int own, bord; comp_rows(c, own, bord);
            if ((own > 8 || own + bord > 16) && own + bord > worst_rows) { worst = c; worst_rows = own + bord; }
        }
        if (worst < 0) break;
        int best = -1, best_deg = -1;
        for (int i = 0; i < nd; i++) {
            if (comp[i] != worst) continue;
            int deg = 0;
            for (int j = 0; j < nd; j++) deg += (j != i && comp[j] == worst && adj(i, j)) ? 1 : 0;
            if (deg > best_deg) { best_deg = deg; best = i; }
        }
        border[best] = 1;
        int nb_total = 0;
        for (int i = 0; i < nf; i++) nb_total += border[i];
        if (nb_total > 16) return P;
    }
    int T = 0;
    for (int i = 0; i < nf; i++) T += border[i];
    if (ncomp < 2 || T > 16 || T < 1) return P;     // nothing to gain without at least two independent blocks
    // components -> at most four groups (a group may hold several components: they simply do not couple), largest first into the
    // group that stays smallest, subject to own <= 8 and own + union of touched border <= 16
    struct Grp { std::vector<int> own; std::vector<char> touch; };
    std::vector<int> order(ncomp);
    std::vector<int> csize(ncomp, 0);
    for (int i = 0; i < nd; i++) if (comp[i] >= 0) csize[comp[i]]++;
    for (int c = 0; c < ncomp; c++) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return csize[a] > csize[b]; });
    const int G = std::min(ncomp, 4);
    std::vector<Grp> grp(G);
    for (auto &q : grp) q.touch.assign(nf, 0);
    for (int c : order) {
        int best = -1, best_rows = 1 << 30;
        for (int gi = 0; gi < G; gi++) {
            std::vector<char> touch = grp[gi].touch;
            int own = (int)grp[gi].own.size() + csize[c], bord = 0;
            for (int b = 0; b < nf; b++) {
                if (!border[b]) continue;
                for (int i = 0; i < nd && !touch[b]; i++) if (comp[i] == c && adj(i, b)) touch[b] = 1;
                bord += touch[b];
            }
            if (own <= 8 && own + bord <= 16 && own + bord < best_rows) { best_rows = own + bord; best = gi; }
        }
        if (best < 0) return P;
        for (int i = 0; i < nd; i++) if (comp[i] == c) grp[best].own.push_back(i);
        for (int b = 0; b < nf; b++) {
            if (!border[b] || grp[best].touch[b]) continue;
            for (int i = 0; i < nd; i++) if (comp[i] == c && adj(i, b)) { grp[best].touch[b] = 1; break; }
        }
    }
    // trailing order: configs (ascending), then constraints (ascending)
    std::vector<int> tindex(nf, -1);
    int t = 0;
    for (int i = 0; i < nf; i++) if (border[i]) { P.tvar[t] = i; tindex[i] = t++; }
    int NG = 0, NB = 0;
    for (auto &q : grp) {
        std::sort(q.own.begin(), q.own.end());
        int bord = 0;
        for (int b = 0; b < nf; b++) bord += q.touch[b];
        NG = std::max(NG, (int)q.own.size()); NB = std::max(NB, bord);
    }
    if (NG + NB > 16 || NG < 1) return P;
    for (int gi = 0; gi < G; gi++) {
        const Grp &q = grp[gi];
        std::vector<int> bl;
        for (int b = 0; b < nf; b++) if (q.touch[b]) bl.push_back(b);   // ascending image index = ascending trailing index
        for (int r = 0; r < 16; r++) {
            int row = -1, trow = -1;
            if (r < NG) row = r < (int)q.own.size() ? q.own[r] : -1;
            else if (r - NG < (int)bl.size()) { row = bl[r - NG]; trow = tindex[row]; }
            P.tab[16 * gi + r] |= (row + 1) | ((trow + 1) << 8);
            P.tab[64 + 16 * gi + r] = (row + 1) | ((trow + 1) << 8);    // column j of the group = the variable of its row j
        }
    }
    for (int i = 0; i < T; i++) P.tab[i] |= (P.tvar[i] + 1) << 16;
    P.ok = 1; P.g = G; P.ng = NG; P.nb = NB; P.t = T;
    return P;
}

// ---- the image in the solver's own order ("packed": round 5) -----------------------------------------------------------------------
// The solver reads the dense image [nf][ld] by gathered 8-byte loads -- per entry a table field, an address and two selects (padding rows,
// border rows' untouched columns): a fifth of its vector instructions.  In the packed image every lane's row [own | border | rhs] is one
// run of 16-byte-aligned doubles, padded to NC2, group g row r at (g * NR + r) * NC2 (NR = ng + nb; four groups are always laid out), the
// trailing system's row i at TB + i * TC2 (16 rows: rows t .. 15 stay zero), the solution vector XS [nf] wherever the caller has nf
// doubles (offset `xs` from the image; the rollout kernels use the dense solvers' scale vector).  Zero is what the
// writer's clear leaves; the identity entries of padding rows are listed in `ones`.  Every structural entry of the dense matrix has
// exactly one place: map[i * (nf + 1) + j] (j = nf: the right-hand side), -1 where the plan has none (an entry that must be zero).
struct BbdPacked {
    int ok = 0, nr = 0, nc2 = 0, tb = 0, tc2 = 0, xs = 0, size = 0;
    std::vector<int> map, ones;
};
inline BbdPacked bbd_pack(const BbdPlan &P, int nf) {
    BbdPacked K;
    if (!P.ok) return K;
    const int NG = P.ng, NB = P.nb, T = P.t;
    // row strides: even (16-byte loads) and not a multiple of four doubles -- at 12 doubles (96 bytes) the 16-byte accesses of lanes eight
    // rows apart meet on the same banks (two-way conflicts: profiles/r04_lds_conflicts.txt), at 14 they do not
    auto stride = [](int n) { int s = (n + 1) & ~1; if (s % 4 == 0) s += 2; return s; };
    K.nr = NG + NB; K.nc2 = stride(NG + NB + 1); K.tb = 4 * K.nr * K.nc2; K.tc2 = stride(T + 1);
    K.xs = K.tb + (T + 1) * K.tc2; K.size = (K.xs + 1) & ~1;        // (trailing rows 0 .. t-1 and one zero row for the lanes past them; the caller places XS)
    K.map.assign((size_t)nf * (nf + 1), -1);
    std::vector<int> tindex(nf, -1);
    for (int g = 0; g < 4; g++)
        for (int r = 0; r < 16; r++) {
            const int w = P.tab[16 * g + r], row = (w & 0xFF) - 1, trow = ((w >> 8) & 0xFF) - 1;
            if (row >= 0 && trow >= 0) tindex[row] = trow;
        }
    for (int i = 0; i < T; i++) tindex[P.tvar[i]] = i;
    for (int g = 0; g < 4; g++) {
        // column j of group g holds variable colvar[j] (own 0 .. NG-1, border NG .. NG+NB-1)
        int colvar[16];
        for (int j = 0; j < 16; j++) colvar[j] = (P.tab[64 + 16 * g + j] & 0xFF) - 1;
        for (int r = 0; r < K.nr; r++) {
            const int row = (P.tab[16 * g + r] & 0xFF) - 1, base = (g * K.nr + r) * K.nc2;
            if (row < 0) { if (r < NG) K.ones.push_back(base + r); continue; }      // a padding own row: identity
            if (r < NG) {       // an own row: every column
                for (int j = 0; j < K.nr; j++) if (colvar[j] >= 0) K.map[(size_t)row * (nf + 1) + colvar[j]] = base + j;
                K.map[(size_t)row * (nf + 1) + nf] = base + K.nr;
            } else {            // a border row of the group: its own columns only (the rest accumulates the Schur update from zero)
                for (int j = 0; j < NG; j++) if (colvar[j] >= 0) K.map[(size_t)row * (nf + 1) + colvar[j]] = base + j;
            }
        }
    }
    for (int i = 0; i < T; i++) {
        for (int j = 0; j < T; j++) K.map[(size_t)P.tvar[i] * (nf + 1) + P.tvar[j]] = K.tb + i * K.tc2 + j;
        K.map[(size_t)P.tvar[i] * (nf + 1) + nf] = K.tb + i * K.tc2 + T;
    }
    if (K.ones.empty()) K.ones.push_back(-1);
    K.ok = 1;
    return K;
}

}  // namespace tg

