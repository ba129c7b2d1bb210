// program.hpp -- host-side "topology compiler": tg_system_desc -> flat device schedule.
//
// The reference walks its frame tree recursively and keeps per-frame tables of 4x4 derivative
// matrices (frame.c:839-2193).  The device path uses an algebraically equal formulation that needs
// no such tables: world poses of the *variable* frames ("joints"), body Jacobians
// J_{F,k} = Ad_{g_F^-1} s_k of every massive frame F w.r.t. every config k on its path, and Lie
// brackets of those 6-vectors for all higher derivatives.  This file flattens the tree into the
// index tables that formulation needs: joints sorted by depth level with constant pre-transforms
// (chains of fixed frames are multiplied out here, once), bodies, (body, path-config) items,
// (item, item) pairs, constraint end points and constraint-Jacobian items.
#pragma once
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/trep_amd.h"
#include "bbd.hpp"

namespace tg {

// Plain-old-data view handed to the kernels (device pointers when on the GPU).
struct DevProg {
    int nq, nd, nk, nu, nc, nf, nX;
    int n_joints, n_levels, n_bodies, n_items, n_pairs, n_endpoints, n_dh, n_cf, n_cfgitems;
    double grav[3];
    const int *level_off;     // [n_levels+1] offsets into lvl_joints
    const int *lvl_joints;    // joints sorted by level (generic level-by-level sweep)
    // chain sweep: joints are numbered chain by chain; round r holds the chains round_off[r]..round_off[r+1]-1, every
    // one of which hangs off a joint of an earlier round (or the world)
    const int *round_off, *ch_first, *ch_len, *ch_parent;
    int n_rounds, n_chains;
    const int *j_parent;      // [n_joints] parent joint or -1
    const int *j_kind;        // [n_joints] TG_TX..TG_RZ
    const int *j_cfg;         // [n_joints]
    const int *j_pre_ident;   // [n_joints] 1 if the constant pre-transform is the identity
    const double *j_pre;      // [n_joints*12]
    const double *jcoef;      // [n_joints*16*4]: local transform entry e = A + B cos q + C sin q + D q
    const double *j_prm;      // [n_joints*12]: the constant pre-transform of joint j with its columns in the order (axis a, a + 1, a + 2, translation)
    const int *b_anchor;      // [n_bodies] joint or -1
    const double *b_C;        // [n_bodies*12]
    const double *b_inertia;  // [n_bodies*4]
    const int *b_item_off;    // [n_bodies+1]
    const int *b_pair_off;    // [n_bodies+1]
    const int *it_body;       // [n_items]
    const int *it_joint;      // [n_items]
    const int *it_cfg;        // [n_items]
    const int *pair_a;        // [n_pairs] item index of the config nearer the root
    const int *pair_b;        // [n_pairs] item index of the config nearer the body (may equal pair_a)
    const int *pair4;         // [n_npairs*4] dynamic-dynamic pairs only: {item a, item b, config a | config b << 16, body}
    int n_npairs;
    // second-derivative kernel: every ORDERED (item x, item y) pair of every body {x, y, config x | config y << 16, body}
    // and every (constraint, dh-item a <= dh-item b) pair {c, na, nb, config a | config b << 16}
    const int *tri4, *cpair4;
    // packed records (one 16-byte load instead of chained look-ups):
    //   it_pack[it]  = {body, 12*joint, joint kind, config | residual slot << 16}
    //   dh_pack[2n]  = {constraint, config, 12*joint (or -1), side | joint kind << 8 | constraint type << 16 | component << 24}
    //   dh_pack[2n+1]= {3*end point 1, 3*end point 2, length config (or -1), 0}
    const int *it_pack, *dh_pack;
    // rows of the attach phase as the lanes read them (eval_both_tab, fetch_attach): at_d[4 i] = {C_b[e], C_b[c], C_b[4 + c], C_b[8 + c]} of body
    // entry i = 12 b + e (c = e & 3), at_i[i] = the body's anchor joint; ae_d[4 i] = {o_0, o_1, o_2, o_r} of end-point coordinate i = 3 e + r,
    // ae_i[i] = its anchor joint -- one vector load per row instead of four scalar ones behind a division by 12
    const double *at_d, *ae_d;
    const int *at_i, *ae_i;
    // the dh items the ROLLOUT needs -- those of the dynamic configs (the residual and the Newton matrix never touch Dh with respect to a
    // kinematic config) -- in dh_pack's layout, word 7 = the item's index n (where its value goes in Dh1 / Dh2); n_dhr of them
    const int *dhr_pack;
    int n_dhr;
    // second-derivative kernel, constraint part: for constraint c and end point E (0/1) the dh items whose joint lies on
    // the end point's path, sorted root-first: cpath_items[cpath_off[2c+E] .. cpath_off[2c+E+1]); dh_pos[2n+E] = position
    // of dh item n in that list or -1.  o_cps: LDS scratch (6 doubles per listed item), aliased with the dead J/W area.
    const int *cpath_off, *cpath_items, *dh_pos;
    int n_cpath, o_cps;
    // body part: bodies are taken in chunks whose per-item prefix / suffix sums (18 doubles per item + 18 per body) fit
    // the dead pose area (o_tps, tps_cap doubles): chunk ci = bodies tchunk[ci] .. tchunk[ci+1]-1; tri_off[b] = first
    // entry of body b in tri4.  n_tchunk = 0: fall back to the direct triple loop.
    const int *tchunk, *tri_off;
    int n_tchunk, o_tps;
    int n_tri, n_cpair;
    // helper-wave kernels (two wavefronts per trajectory, mvi_core.hpp): the same pair lists split in two parts that never
    // accumulate into the same table entry -- every pair belongs to the part of its UNORDERED CONFIG PAIR, parts balanced,
    // order inside a part as in the original list -- so that the two waves' LDS atomics never meet at an address and the
    // sums keep a fixed order:  wp_a / wp_b [n_pairs] (all (item, item) pairs; part 0 = [0, wp_split)),  wt_a / wt_b
    // [n_pairs] (the same per chunk of bodies: chunk ci occupies the range of pair_a, part 0 ends at wt_split[ci]),
    // wcp4 [n_cpair * 4] (constraint pairs, part 0 = [0, wc_split)).
    const int *wp_a, *wp_b, *wt_a, *wt_b, *wt_split, *wcp4;
    int wp_split, wc_split;
    const int *cfg_item_off;  // [nq+1] CSR config -> items
    const int *cfg_items;     // [n_items]
    const int *it_slot;       // [n_items] position of the item in the config-sorted order (inverse of cfg_items)
    const int *e_anchor;      // [n_endpoints]
    const double *e_off;      // [n_endpoints*3]
    const int *c_type, *c_e1, *c_e2, *c_cfg, *c_comp;
    const double *c_dist, *c_tol;
    const int *dh_lookup;     // [nc*nq] index into the dh items of (constraint, config), -1 if independent
    const int *dh_c, *dh_cfg, *dh_joint, *dh_side; // side: bit0 on e1's path, bit1 on e2's path, bit2 length config
    const double *damp;       // [nd] summed damping coefficients
    const int *cf_cfg, *cf_in;
    // LDS layout (offsets in doubles from the team's base)
    int o_q1, o_q2, o_p1, o_lam, o_u, o_dq, o_f, o_sc, o_G, o_gB, o_pE, o_J, o_W, o_vB, o_gam, o_Ldq, o_Lddq,
        o_Dh1, o_Dh2, o_Df, o_scal, o_misc, o_dqi, o_nu, o_sched, o_I, o_ctol, gjc_ok;
    int tab_ok;               // 1: the per-lane table rows of the rollout's Newton iteration fit two trips of a wavefront (mvi_core.hpp eval_both_tab)
    int sched_ok;             // 1 / 2: every round has <= 16 / <= 8 chains: the chain schedule is staged in LDS (o_sched)
    int df_ld;
    int dh_ld;                // 0: the step kernel keeps Dh1/Dh2 compact (one value per dh item)
    int lds_per_team;
    // first-derivative kernel: extra arrays appended after the step layout
    int d_o_Dh1, d_o_Dh2, d_o_AUG, d_aug_ld, d_o_T12, d_o_T22, d_nrhs, d_lds_per_team;
    // z-contracted second-derivative kernel: contracted Hessians H11/H12/H22 [nq][nq], G1 [nq][nc], vectors
    const int *cu_off;        // [nc+1] dh items of each constraint (its dependent configs)
    // spring potentials, kept near the end: the spring-free kernels' argument layout stays what it was
    const double *cs_k, *cs_kq0;  // [nq] config springs: sum k and sum k q0 per config (V_dq = cs_k q - cs_kq0)
    const double *s_k, *s_x0;     // [n_springs]
    int n_springs, n_sdh, n_spair, o_sV, o_sH, has_cs;   // two-point springs: dh items / pairs follow the constraints' in the same tables
    // continuous-dynamics derivative kernel (MODE_DYN_DERIV1): KKT matrix + one column per derivative variable, prefix vectors
    int g_nrhs, g_ld, g_o_AUG, g_o_P, g_o_X, g_o_aF, g_o_x, g_lds_per_team, g_max_cu;
    // plane constraints (TG_CONSTRAINT_PLANE): normal in the coordinates of the plane frame's anchor joint; world normal in LDS
    const double *c_nloc;   // [3 * (nc + n_springs + n_wrenches)]
    int has_plane, o_nE;
    // point forces (HybridWrench, force part): items / pairs follow the springs' in the dh / cpair tables
    const int *wr_in; const double *wr_const;   // [6 * n_wrenches] input index or -1, constant component (fx fy fz tx ty tz)
    const int *wr_kind;                         // [n_wrenches] 0 hybrid, 1 spatial, 2 body
    const double *wr_Rloc; int o_wR;            // [9 * n_wrenches] frame rotation relative to its anchor joint; world rotation in LDS
    int n_wrenches, n_wdh, n_wpair, o_wF, o_wH, o_wD, e_o_wT, e_o_Hu;
    // linear dampers (spring elements with a coefficient c): d|p1-p2|/dq per item, its q-derivative per pair, rate per element
    const double *s_c; int has_damper, o_sX, o_sVq, o_sXX, o_svel, o_sF;
    int n_true_springs, e_o_sT, e_o_sWX, e_o_sWXq;   // second derivatives: per pair (P, R), per element (sum w x, sum w), per item sum_o w_o x_ob
    const int *tab_i; const double *tab_d; int n_tab_i, n_tab_d;  // the packed table buffers (all pointers above point into them)
    int e_o_H11, e_o_H12, e_o_H22, e_o_G1, e_o_w, e_o_zq, e_o_zp, e_o_vec, e_o_vec2, e_lds_per_team;
    const double *cs_c0;          // [nq] 1/2 sum k q0^2 (constant part of V)
    // NonlinearConfigSpring: per spring (config, first table row, pieces), (m, b); table rows (left knot, a, b, c, d, e, f) per piece
    const int *ncs_i; const double *ncs_mb, *ncs_tab; int n_ncs;
    // structured Newton solve of the system-specialised rollout kernels (bbd.hpp): groups, largest own block, largest border list,
    // trailing size; the plan tables (bbd_tab [128], bbd_tvar [16]) and the LDS offset they are staged at (rollout kernels only:
    // the slice's last 64 doubles, which the derivative layouts overlay)
    // composite form of the Newton matrix's inertial part (system-specialised rollout kernels, mvi_core.hpp newton_matrix_composite): the
    // (body, item, item) pairs of a config pair (a, b) all share the bodies below b, so their sum is a bilinear form of the WORLD-frame
    // joint twists in the composite inertia / momentum of that subtree.  cmp_rep [nd]: an item of config k (or -1); cmp_grp [nd]: the
    // subtree group of config k (configs with the same set of bodies below them share one); cmp_goff / cmp_gbody: the bodies of each
    // group; cmp_pair [n_cmpairs]: a | b << 16 for every dynamic config pair with a on the path to b; LDS (inside the J / W area, dead
    // once the residual is formed): o_cmp 16 doubles per group, o_csw 12 per config (s, w), o_ccz 15 per config (I s, Z, G x g)
    int cmp_ok, n_cgroups, n_cmpairs, o_cmp, o_csw, o_ccz, o_cmpt;   // o_cmpt: per config item | body << 12 | group << 20 (ints), behind the plan tables (the per-body world entries go to the dead joint-pose area)
    int cmp_gmask[32];        // bit F of cmp_gmask[g]: body F belongs to subtree group g
    const int *cmp_rep, *cmp_grp, *cmp_goff, *cmp_gbody, *cmp_pair;
    // quad-lane chain sweep of the specialised dual pose sweep (mvi_core.hpp, chain_round_quads): a round's 2 x chains instances (chain slot
    // i / 2 of pose set i % 2) go five to a pass (twelve lanes = the 3 x 4 entries of the running pose each); sw_np[r] passes in round r,
    // sw_len[4 r + p] the longest chain of pass p (the unrolled trip count), sw_maxlen the longest of all
    int sw_ok, sw_maxlen;
    int sw_np[4];
    int sw_len[16];
    int sw_inst[80];          // instance q of pass p of round r, [(4 r + p) * 5 + q]: chain slot | pose set << 8 | 1 << 9 (0: none)
    // the (pose set, joint) items the ROLLOUT's dual sweep needs -- midpoint poses of the joints that carry a body, q2 poses of the joints
    // on the way to a constraint end point -- rotary ones first: config | kind << 12 | joint << 16 | pose set << 28 (sj_list), n_sj of them,
    // the first n_sj_rot rotary
    int n_sj, n_sj_rot;
    const int *sj_list;
    const int *sj_full;       // the same for EVERY joint in both pose sets (2 n_joints items, the first 2 x rotary ones rotary): the derivative kernels' dual sweep
    int max_cfg_items;        // most items any dynamic config has (= bodies below it): trip count of the specialised residual sum
    int bbd_ok, bbd_g, bbd_ng, bbd_nb, bbd_t, o_bbd;
    int bbd_tvar[16];         // image index of trailing variable i
    const int *bbd_tab;
    // world-frame evaluation of the rollout's residual (system-specialised kernels, mvi_core.hpp eval_world): the Lagrangian terms of
    // system.c:129-202 per CONFIG from world-frame joint twists and the composite momentum of the subtree below it, instead of per
    // (body, config) item.  wev_lane [64][4]: lane l < nd is config l -- words 0..2 = its proper ancestors on the path, root first, one
    // byte each, padded with nd (the all-zero record), word 3 = 12 * joint | kind << 16 | subtree group << 24; lane nd + 3 b + r is
    // (body b, axis r) -- words 0..2 = the configs of the body's path, word 3 = b | r << 8.  wev_depth: longest list.
    int wev_ok, wev_depth, wev_rc_ident;      // wev_rc_ident: every body's constant offset from its anchor joint is a pure translation
    const int *wev_lane;
    // the Newton image in the structured solve's own order (bbd.hpp, BbdPacked; kernels with the world-frame evaluation): sizes, the dense
    // -> packed map [nf * (nf + 1)] (test hook), the addresses of the identity entries (bbd_pk_nones of them), and the writers' tables:
    // wev_pairx [n_cmpairs] a | b << 6 | packed address of (a, b) << 12 | of (b, a) << 22;  wev_dhx [n_dhr] Dh item n | address of
    // (config, constraint) << 8 | of (constraint, config) << 18;  the right-hand side of variable i at bbd_map[i * (nf + 1) + nf]
    int bbd_pk_ok, bbd_pk_nr, bbd_pk_nc2, bbd_pk_tb, bbd_pk_tc2, bbd_pk_xs, bbd_pk_size, bbd_pk_nones;
    const int *bbd_map, *bbd_ones, *wev_pairx, *wev_dhx;
    // COMPACT slice of the first-derivative kernel (MODE_DERIV1 only; a_ok): with the step layout's pose union [o_sc, end of the base region)
    // dead once the midpoint is evaluated, and the two D.D2L2 tables only filled by the (item, item) pair loop after that, the table T12 moves
    // INTO the pose union, the full-width constraint Jacobians Dh1 / Dh2 (dead after the KKT matrix's constant blocks) into the place T22 takes
    // later, and the KKT image loses the nc unit columns only the second-derivative kernel appends: puppet 63.2 -> 51.5 KB per trajectory, i.e.
    // three resident workgroups per CU instead of two (measured with timing mocks first: -28 % kernel time).  The tables are cleared late
    // (after the constant blocks) in this layout; the second-derivative kernel keeps the d_* / e_* layout.
    int a_ok, a_o_T12, a_o_AUG, a_aug_ld, a_o_T22, a_lds_per_team;
};

struct HostProgram {
    DevProg p{};  // sizes + LDS layout filled; pointers left null (set by the owner)
    std::vector<int> level_off, lvl_joints, round_off, ch_first, ch_len, ch_parent, j_parent, j_kind, j_cfg, j_pre_ident;
    std::vector<double> j_pre;
    std::vector<int> b_anchor;
    std::vector<double> b_C, b_inertia;
    std::vector<int> b_item_off, b_pair_off, it_body, it_joint, it_cfg, pair_a, pair_b, cfg_item_off, cfg_items, it_slot, pair4, tri4, cpair4, it_pack, dh_pack, cpath_off, cpath_items, dh_pos, tchunk, tri_off, dhr_pack, at_i, ae_i, sj_list, sj_full;
    std::vector<int> wp_a, wp_b, wt_a, wt_b, wt_split, wcp4;
    std::vector<int> e_anchor;
    std::vector<double> e_off;
    std::vector<int> c_type, c_e1, c_e2, c_cfg, c_comp;
    std::vector<double> c_dist, c_tol;
    std::vector<int> dh_c, dh_cfg, dh_joint, dh_side, dh_lookup, cu_off;
    std::vector<double> damp, cs_k, cs_kq0, cs_c0, s_k, s_x0, s_c, c_nloc, wr_const, wr_Rloc, ncs_mb, ncs_tab;
    std::vector<int> ncs_i, bbd_tab, cmp_rep, cmp_grp, cmp_goff, cmp_gbody, cmp_pair, wev_lane, bbd_map, bbd_ones, wev_pairx, wev_dhx;
    std::vector<unsigned char> newton_pattern;   // [nf * nf] structural non-zeros of the Newton matrix (symmetrised), host side only
    std::vector<int> wr_in, wr_kind;
    std::vector<int> cf_cfg, cf_in;
    std::vector<double> jcoef;      // [n_joints*16*4] local-transform coefficients (see pose_sweep)
    std::vector<double> at_d, ae_d;   // attach-phase rows (DevProg::at_d / ae_d)
    std::vector<double> j_prm;      // [n_joints*12] pre-transform, columns permuted to (axis, next, next-but-one, translation) (see pose_sweep_dual)
    int max_depth = 0;
    // all tables packed into two pools; bind() points a DevProg's table pointers into (copies of) them
    std::vector<int> ipool;
    std::vector<double> dpool;
    std::vector<size_t> ioff, doff;
    void pack();
    void bind(DevProg &P, const int *I, const double *D) const;
};

namespace detail {
struct M34 {  // row-major 3x4 rigid transform
    double m[12];
};
inline M34 ident() {
    M34 r{};
    r.m[0] = r.m[5] = r.m[10] = 1.0;
    return r;
}
inline M34 mul(const M34 &a, const M34 &b) {
    M34 r{};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double s = (j == 3) ? a.m[4 * i + 3] : 0.0;
            for (int k = 0; k < 3; k++) s += a.m[4 * i + k] * b.m[4 * k + j];
            r.m[4 * i + j] = s;
        }
    return r;
}
inline bool is_ident(const M34 &a) {
    M34 i = ident();
    return std::memcmp(a.m, i.m, sizeof(a.m)) == 0;
}
// local transform of a FIXED frame (frame.c:839-1068 with x = frame->value, or the stored SE(3))
inline M34 fixed_local(const tg_system_desc *d, int f) {
    int t = d->frame_transform[f];
    M34 r = ident();
    double x = d->frame_value[f];
    if (t == TG_CONST_SE3) {
        std::memcpy(r.m, d->frame_lg + 12 * (size_t)f, sizeof(r.m));
    } else if (t == TG_TX || t == TG_TY || t == TG_TZ) {
        r.m[4 * (t - TG_TX) + 3] = x;
    } else if (t == TG_RX) {
        r.m[5] = std::cos(x); r.m[6] = -std::sin(x); r.m[9] = std::sin(x); r.m[10] = std::cos(x);
    } else if (t == TG_RY) {
        r.m[0] = std::cos(x); r.m[2] = std::sin(x); r.m[8] = -std::sin(x); r.m[10] = std::cos(x);
    } else if (t == TG_RZ) {
        r.m[0] = std::cos(x); r.m[1] = -std::sin(x); r.m[4] = std::sin(x); r.m[5] = std::cos(x);
    }
    return r;
}
}  // namespace detail

inline HostProgram build_program(const tg_system_desc *d) {
    using namespace detail;
    HostProgram H;
    const int nfr = d->n_frames, nq = d->n_configs, nd = d->n_dyn, nc = d->n_constraints;
    if (nd + d->n_kin != nq) throw std::runtime_error("n_dyn + n_kin != n_configs");
    for (int f = 0; f < nfr; f++) {
        int t = d->frame_transform[f];
        if (t < TG_WORLD || t > TG_CONST_SE3) throw std::runtime_error("unknown frame transform");
        if (d->frame_config[f] >= 0 && (t < TG_TX || t > TG_RZ)) throw std::runtime_error("config on a non-parametric frame");
        if (f > 0 && (d->frame_parent[f] < 0 || d->frame_parent[f] >= f)) throw std::runtime_error("frames must be listed parent-first");
    }
    // joints = variable frames.  Numbering: CHAIN ORDER.  A chain is a maximal run of joints in which every joint is
    // the only joint-child of its predecessor; round 0 holds the chains that hang off the world, round r+1 the chains
    // that hang off a joint of round r.  The device sweeps one chain per lane group, joint after joint in registers,
    // with one barrier per ROUND instead of one per tree level (puppet: 2 rounds for 11 levels).
    std::vector<int> jframes;
    int n_levels = 0;
    for (int f = 0; f < nfr; f++)
        if (d->frame_config[f] >= 0) {
            jframes.push_back(f);
            n_levels = std::max(n_levels, d->config_gen[d->frame_config[f]] + 1);
        }
    std::vector<int> order;
    {
        std::vector<int> vparent(nfr, -1);            // nearest variable proper ancestor frame
        std::vector<int> nearest(nfr, -1);            // nearest variable ancestor-or-self
        for (int f = 1; f < nfr; f++) {
            const int p = d->frame_parent[f];
            vparent[f] = nearest[p];
            nearest[f] = d->frame_config[f] >= 0 ? f : nearest[p];
        }
        std::vector<std::vector<int>> kids(nfr);
        std::vector<int> roots;
        for (int f : jframes) { if (vparent[f] < 0) roots.push_back(f); else kids[vparent[f]].push_back(f); }
        std::vector<int> starts = roots;
        H.round_off.push_back(0);
        while (!starts.empty()) {
            std::vector<int> next;
            for (int f0 : starts) {
                H.ch_first.push_back((int)order.size());
                int f = f0, len = 0;
                for (;;) {
                    order.push_back(f); len++;
                    if (kids[f].size() != 1) break;
                    f = kids[f][0];
                }
                H.ch_len.push_back(len);
                for (int c : kids[f]) next.push_back(c);
            }
            H.round_off.push_back((int)H.ch_first.size());
            starts.swap(next);
        }
        if (order.size() != jframes.size()) throw std::runtime_error("joint tree traversal error");
    }
    std::vector<int> joint_of_frame(nfr, -1), joint_of_cfg(nq, -1);
    for (size_t j = 0; j < order.size(); j++) {
        joint_of_frame[order[j]] = (int)j;
        joint_of_cfg[d->frame_config[order[j]]] = (int)j;
    }
    // anchor (nearest variable ancestor-or-self) and constant offset from it, for every frame
    std::vector<int> anchor(nfr, -1);
    std::vector<M34> offset(nfr, ident());
    for (int f = 1; f < nfr; f++) {
        if (d->frame_config[f] >= 0) {
            anchor[f] = joint_of_frame[f];
            offset[f] = ident();
        } else {
            int p = d->frame_parent[f];
            anchor[f] = anchor[p];
            offset[f] = mul(offset[p], fixed_local(d, f));
        }
    }
    const int nj = (int)order.size();
    H.j_parent.resize(nj); H.j_kind.resize(nj); H.j_cfg.resize(nj); H.j_pre_ident.resize(nj); H.j_pre.resize(12 * (size_t)nj);
    for (int j = 0; j < nj; j++) {
        int f = order[j], p = d->frame_parent[f];
        H.j_parent[j] = anchor[p];
        H.j_kind[j] = d->frame_transform[f];
        H.j_cfg[j] = d->frame_config[f];
        H.j_pre_ident[j] = is_ident(offset[p]) ? 1 : 0;
        std::memcpy(&H.j_pre[12 * (size_t)j], offset[p].m, sizeof(offset[p].m));
        if (H.j_parent[j] >= 0 && H.j_parent[j] >= j) throw std::runtime_error("joint ordering error");
    }
    for (size_t c = 0; c < H.ch_first.size(); c++) H.ch_parent.push_back(H.j_parent[H.ch_first[c]]);
    H.level_off.assign(n_levels + 1, 0);
    for (int L = 0; L < n_levels; L++) {
        for (int j = 0; j < nj; j++)
            if (d->config_gen[H.j_cfg[j]] == L) H.lvl_joints.push_back(j);
        H.level_off[L + 1] = (int)H.lvl_joints.size();
    }
    // local transform of joint j as a function of its coordinate: entry e (row l, column c) of
    // pre_j * lg(q) = A + B*s1 + C*s0 with (s0, s1) = (sin q, cos q) for a rotary joint and (q, 0) for a
    // prismatic one -- removes every kind/identity branch from the device sweep.
    H.jcoef.assign(16 * 4 * (size_t)nj, 0.0);
    for (int j = 0; j < nj; j++) {
        const double *pre = &H.j_pre[12 * (size_t)j];
        const int kind = H.j_kind[j];
        for (int l = 0; l < 3; l++)
            for (int c = 0; c < 4; c++) {
                double *k = &H.jcoef[((size_t)j * 16 + 4 * l + c) * 4];
                if (kind <= TG_TZ) {
                    k[0] = pre[4 * l + c];
                    if (c == 3) k[2] = pre[4 * l + (kind - TG_TX)];
                } else {
                    const int a = kind - TG_RX, b = (a + 1) % 3, cc = (a + 2) % 3;
                    if (c == b) { k[1] = pre[4 * l + b]; k[2] = pre[4 * l + cc]; }
                    else if (c == cc) { k[1] = pre[4 * l + cc]; k[2] = -pre[4 * l + b]; }
                    else k[0] = pre[4 * l + c];
                }
            }
    }
    H.j_prm.assign(12 * (size_t)nj, 0.0);
    for (int j = 0; j < nj; j++) {
        const double *pre = &H.j_pre[12 * (size_t)j];
        const int kind = H.j_kind[j], a = kind <= TG_TZ ? kind - TG_TX : kind - TG_RX;
        for (int l = 0; l < 3; l++) {
            double *o = &H.j_prm[12 * (size_t)j + 4 * l];
            o[0] = pre[4 * l + a]; o[1] = pre[4 * l + (a + 1) % 3]; o[2] = pre[4 * l + (a + 2) % 3]; o[3] = pre[4 * l + 3];
        }
    }
    // bodies and (body, path config) items
    const int nb = d->n_masses;
    H.b_anchor.resize(nb); H.b_C.resize(12 * (size_t)nb); H.b_inertia.resize(4 * (size_t)nb);
    H.b_item_off.assign(nb + 1, 0); H.b_pair_off.assign(nb + 1, 0);
    for (int b = 0; b < nb; b++) {
        int f = d->masses[b];
        H.b_anchor[b] = anchor[f];
        std::memcpy(&H.b_C[12 * (size_t)b], offset[f].m, sizeof(offset[f].m));
        std::memcpy(&H.b_inertia[4 * (size_t)b], d->frame_inertia + 4 * (size_t)f, 4 * sizeof(double));
        int n = d->frame_cache_size[f];
        H.max_depth = std::max(H.max_depth, n);
        int first = (int)H.it_body.size();
        for (int i = 0; i < n; i++) {
            int c = d->frame_cache_index[(size_t)f * (nq + 1) + i];
            if (c < 0 || joint_of_cfg[c] < 0) throw std::runtime_error("path config without a frame");
            H.it_body.push_back(b); H.it_joint.push_back(joint_of_cfg[c]); H.it_cfg.push_back(c);
        }
        H.b_item_off[b + 1] = (int)H.it_body.size();
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) { H.pair_a.push_back(first + i); H.pair_b.push_back(first + j); }
        H.b_pair_off[b + 1] = (int)H.pair_a.size();
    }
    const int nitems = (int)H.it_body.size();
    H.cfg_item_off.assign(nq + 1, 0);
    for (int c = 0; c < nq; c++) {
        for (int it = 0; it < nitems; it++)
            if (H.it_cfg[it] == c) H.cfg_items.push_back(it);
        H.cfg_item_off[c + 1] = (int)H.cfg_items.size();
    }
    for (size_t n = 0; n < H.pair_a.size(); n++) {   // the Newton matrix only needs dynamic-dynamic pairs
        const int ca = H.it_cfg[H.pair_a[n]], cb = H.it_cfg[H.pair_b[n]];
        if (ca >= nd || cb >= nd) continue;
        if (ca >= 65536 || cb >= 32768) throw std::runtime_error("too many configs for the packed pair table");
        H.pair4.push_back(H.pair_a[n]); H.pair4.push_back(H.pair_b[n]);
        H.pair4.push_back(ca | (cb << 16)); H.pair4.push_back(H.it_body[H.pair_a[n]]);
    }
    H.it_slot.assign(nitems, 0);
    for (int n = 0; n < nitems; n++) H.it_slot[H.cfg_items[n]] = n;
    for (int n = 0; n < nitems; n++) {
        if (H.it_cfg[n] >= 65536 || H.it_slot[n] >= 32768) throw std::runtime_error("too many items for the packed item table");
        H.it_pack.push_back(H.it_body[n]); H.it_pack.push_back(12 * H.it_joint[n]);
        H.it_pack.push_back(H.j_kind[H.it_joint[n]]); H.it_pack.push_back(H.it_cfg[n] | (H.it_slot[n] << 16));
    }
    for (int b = 0; b <= nb; b++) { const int n0 = b ? H.b_item_off[b] - H.b_item_off[b - 1] : 0; H.tri_off.push_back(b ? H.tri_off[b - 1] + n0 * n0 : 0); }
    for (int b = 0; b < nb; b++)
        for (int x = H.b_item_off[b]; x < H.b_item_off[b + 1]; x++)
            for (int y = H.b_item_off[b]; y < H.b_item_off[b + 1]; y++) {
                H.tri4.push_back(x); H.tri4.push_back(y);
                H.tri4.push_back(H.it_cfg[x] | (H.it_cfg[y] << 16)); H.tri4.push_back(b);
            }
    // constraint end points (unique frames) and constraint-Jacobian items
    std::map<int, int> ep_of_frame;
    auto endpoint = [&](int f) {
        auto it = ep_of_frame.find(f);
        if (it != ep_of_frame.end()) return it->second;
        int e = (int)H.e_anchor.size();
        ep_of_frame[f] = e;
        H.e_anchor.push_back(anchor[f]);
        H.e_off.push_back(offset[f].m[3]); H.e_off.push_back(offset[f].m[7]); H.e_off.push_back(offset[f].m[11]);
        return e;
    };
    H.dh_lookup.assign((size_t)nc * nq, -1);
    bool has_plane = false;
    for (int c = 0; c < nc; c++) {
        int t = d->constraint_type[c];
        if (t != TG_CONSTRAINT_DISTANCE && t != TG_CONSTRAINT_POINT && t != TG_CONSTRAINT_PLANE) throw std::runtime_error("unknown constraint type");
        int f1 = d->constraint_frame1[c], f2 = d->constraint_frame2[c];
        H.c_type.push_back(t); H.c_e1.push_back(endpoint(f1)); H.c_e2.push_back(endpoint(f2));
        int lc = (t == TG_CONSTRAINT_DISTANCE) ? d->constraint_config[c] : -1;
        H.c_cfg.push_back(lc); H.c_comp.push_back(d->constraint_component[c]);
        H.c_dist.push_back(d->constraint_distance[c]); H.c_tol.push_back(d->constraint_tolerance[c]);
        {   // plane normal carried into the coordinates of the plane frame's anchor joint (constant rotation)
            double nl[3] = {0, 0, 0};
            if (t == TG_CONSTRAINT_PLANE) {
                const double *n = d->constraint_normal + 3 * (size_t)c;
                for (int r = 0; r < 3; r++) nl[r] = offset[f1].m[4 * r] * n[0] + offset[f1].m[4 * r + 1] * n[1] + offset[f1].m[4 * r + 2] * n[2];
                has_plane = true;
            }
            H.c_nloc.push_back(nl[0]); H.c_nloc.push_back(nl[1]); H.c_nloc.push_back(nl[2]);
        }
        for (int k = 0; k < nq; k++) {
            int g = d->config_gen[k];
            bool on1 = d->frame_cache_index[(size_t)f1 * (nq + 1) + g] == k;
            bool on2 = d->frame_cache_index[(size_t)f2 * (nq + 1) + g] == k;
            bool isl = (k == lc);
            if (!on1 && !on2 && !isl) continue;
            H.dh_lookup[(size_t)c * nq + k] = (int)H.dh_c.size();
            H.dh_c.push_back(c); H.dh_cfg.push_back(k); H.dh_joint.push_back(joint_of_cfg[k]);
            H.dh_side.push_back((on1 ? 1 : 0) | (on2 ? 2 : 0) | (isl ? 4 : 0));
        }
    }
    // two-point springs (linearspring.c): same end point / dependent-config items as a distance constraint, listed after
    // the constraints' so that every constraint loop (n_dh, n_cpair, nc) leaves them out
    const int n_dh_con = (int)H.dh_c.size();
    const int n_true_springs = d->n_linear_springs;
    const int ns = n_true_springs + d->n_linear_dampers;   // dampers are spring elements with k = 0 and a coefficient c
    for (int s = 0; s < ns; s++) {
        const bool damper = s >= n_true_springs;
        const int f1 = damper ? d->linear_damper_frame1[s - n_true_springs] : d->linear_spring_frame1[s];
        const int f2 = damper ? d->linear_damper_frame2[s - n_true_springs] : d->linear_spring_frame2[s], c = nc + s;
        if (f1 < 0 || f1 >= d->n_frames || f2 < 0 || f2 >= d->n_frames) throw std::runtime_error("linear spring: bad frame index");
        H.c_type.push_back(9); H.c_e1.push_back(endpoint(f1)); H.c_e2.push_back(endpoint(f2));   // 9: not a constraint type
        H.c_nloc.push_back(0.0); H.c_nloc.push_back(0.0); H.c_nloc.push_back(0.0);
        H.c_cfg.push_back(-1); H.c_comp.push_back(0); H.c_dist.push_back(0.0); H.c_tol.push_back(0.0);
        H.s_k.push_back(damper ? 0.0 : d->linear_spring_k[s]); H.s_x0.push_back(damper ? 0.0 : d->linear_spring_x0[s]);
        H.s_c.push_back(damper ? d->linear_damper_c[s - n_true_springs] : 0.0);
        for (int k = 0; k < nq; k++) {
            int g = d->config_gen[k];
            bool on1 = d->frame_cache_index[(size_t)f1 * (nq + 1) + g] == k;
            bool on2 = d->frame_cache_index[(size_t)f2 * (nq + 1) + g] == k;
            if (!on1 && !on2) continue;
            H.dh_c.push_back(c); H.dh_cfg.push_back(k); H.dh_joint.push_back(joint_of_cfg[k]);
            H.dh_side.push_back((on1 ? 1 : 0) | (on2 ? 2 : 0));
        }
    }
    // point forces (hybridwrench.c, force part): one-ended elements listed after the springs
    const int n_dh_spr = (int)H.dh_c.size();
    const int nw = d->n_hybrid_wrenches;
    for (int w = 0; w < nw; w++) {
        const int f1 = d->hybrid_wrench_frame[w], c = nc + ns + w;
        if (f1 < 0 || f1 >= d->n_frames) throw std::runtime_error("hybrid wrench: bad frame index");
        H.c_type.push_back(9); H.c_e1.push_back(endpoint(f1)); H.c_e2.push_back(endpoint(f1));
        H.c_nloc.push_back(0.0); H.c_nloc.push_back(0.0); H.c_nloc.push_back(0.0);
        H.c_cfg.push_back(-1); H.c_comp.push_back(0); H.c_dist.push_back(0.0); H.c_tol.push_back(0.0);
        H.wr_kind.push_back(d->hybrid_wrench_kind[w]);
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) H.wr_Rloc.push_back(offset[f1].m[4 * r + cc]);
        for (int s6 = 0; s6 < 6; s6++) {
            const int in = d->hybrid_wrench_input[6 * w + s6];
            if (in >= d->n_inputs) throw std::runtime_error("hybrid wrench: bad input index");
            H.wr_in.push_back(in); H.wr_const.push_back(d->hybrid_wrench_const[6 * w + s6]);
        }
        for (int k = 0; k < nq; k++) {
            if (d->frame_cache_index[(size_t)f1 * (nq + 1) + d->config_gen[k]] != k) continue;
            H.dh_c.push_back(c); H.dh_cfg.push_back(k); H.dh_joint.push_back(joint_of_cfg[k]); H.dh_side.push_back(1);
        }
    }
    const int nel = nc + ns + nw;
    H.cu_off.assign(nel + 1, 0);
    for (size_t n = 0; n < H.dh_c.size(); n++) H.cu_off[H.dh_c[n] + 1] = (int)n + 1;
    for (int c = 0; c < nel; c++) if (H.cu_off[c + 1] < H.cu_off[c]) H.cu_off[c + 1] = H.cu_off[c];
    H.cpath_off.assign(2 * nc + 1, 0);
    H.dh_pos.assign(2 * H.dh_c.size(), -1);
    for (int c = 0; c < nc; c++)
        for (int E = 0; E < 2; E++) {
            std::vector<std::pair<int, int>> on;   // (joint, dh item)
            for (int n = H.cu_off[c]; n < H.cu_off[c + 1]; n++)
                if (H.dh_joint[n] >= 0 && (H.dh_side[n] & (1 << E))) on.push_back({H.dh_joint[n], n});
            std::sort(on.begin(), on.end());
            for (size_t t = 0; t < on.size(); t++) { H.dh_pos[2 * on[t].second + E] = (int)t; H.cpath_items.push_back(on[t].second); }
            H.cpath_off[2 * c + E + 1] = (int)H.cpath_items.size();
        }
    for (size_t n = 0; n < H.dh_c.size(); n++) {
        const int c = H.dh_c[n], j = H.dh_joint[n];
        H.dh_pack.push_back(c); H.dh_pack.push_back(H.dh_cfg[n]); H.dh_pack.push_back(j >= 0 ? 12 * j : -1);
        H.dh_pack.push_back(H.dh_side[n] | ((j >= 0 ? H.j_kind[j] : 0) << 8) | (H.c_type[c] << 16) | ((H.c_comp[c] & 0xFF) << 24));
        H.dh_pack.push_back(3 * H.c_e1[c]); H.dh_pack.push_back(3 * H.c_e2[c]); H.dh_pack.push_back(H.c_cfg[c]); H.dh_pack.push_back(0);
    }
    for (int i = 0; i < 12 * (int)H.b_anchor.size(); i++) {
        const int b = i / 12, e = i % 12, c = e & 3;
        const double *C = &H.b_C[12 * (size_t)b];
        H.at_d.push_back(C[e]); H.at_d.push_back(C[c]); H.at_d.push_back(C[4 + c]); H.at_d.push_back(C[8 + c]);
        H.at_i.push_back(H.b_anchor[b]);
    }
    for (int i = 0; i < 3 * (int)H.e_anchor.size(); i++) {
        const int e = i / 3, r = i % 3;
        const double *o = &H.e_off[3 * (size_t)e];
        H.ae_d.push_back(o[0]); H.ae_d.push_back(o[1]); H.ae_d.push_back(o[2]); H.ae_d.push_back(o[r]);
        H.ae_i.push_back(H.e_anchor[e]);
    }
    if (H.at_i.empty()) { H.at_i.push_back(-1); H.at_d.assign(4, 0.0); }
    if (H.ae_i.empty()) { H.ae_i.push_back(-1); H.ae_d.assign(4, 0.0); }
    for (int n = 0; n < n_dh_con; n++) {
        if (H.dh_cfg[n] >= nd) continue;
        for (int w = 0; w < 7; w++) H.dhr_pack.push_back(H.dh_pack[8 * (size_t)n + w]);
        H.dhr_pack.push_back(n);
    }
    int n_cpair_con = 0, n_cpair_spr = 0;
    for (int c = 0; c <= nel; c++) {
        if (c == nc) n_cpair_con = (int)(H.cpair4.size() / 4);
        if (c == nc + ns) n_cpair_spr = (int)(H.cpair4.size() / 4);
        if (c == nel) break;
        for (int na = H.cu_off[c]; na < H.cu_off[c + 1]; na++)
            for (int nb2 = na; nb2 < H.cu_off[c + 1]; nb2++) {
                H.cpair4.push_back(c); H.cpair4.push_back(na); H.cpair4.push_back(nb2);
                H.cpair4.push_back(H.dh_cfg[na] | (H.dh_cfg[nb2] << 16));
            }
    }
    // forces / potentials
    H.damp.assign(nd, 0.0);
    for (int i = 0; i < d->n_damping; i++)
        for (int k = 0; k < nd; k++) H.damp[k] += d->damping[(size_t)i * nd + k];
    H.cs_k.assign(nq, 0.0); H.cs_kq0.assign(nq, 0.0); H.cs_c0.assign(nq, 0.0);
    for (int i = 0; i < d->n_config_springs; i++) {
        const int c = d->config_spring_config[i];
        if (c < 0 || c >= nq) throw std::runtime_error("config spring: bad config index");
        H.cs_k[c] += d->config_spring_k[i]; H.cs_kq0[c] += d->config_spring_k[i] * d->config_spring_q0[i];
        H.cs_c0[c] += 0.5 * d->config_spring_k[i] * d->config_spring_q0[i] * d->config_spring_q0[i];
    }
    for (int i = 0, row = 0; i < d->n_nonlinear_springs; i++) {
        const int c = d->nonlinear_spring_config[i], r0 = d->nonlinear_spring_first[i], r1 = d->nonlinear_spring_first[i + 1];
        if (c < 0 || c >= nq || r1 - r0 < 1) throw std::runtime_error("nonlinear config spring: bad config index or empty spline");
        H.ncs_i.push_back(c); H.ncs_i.push_back(row); H.ncs_i.push_back(r1 - r0);
        H.ncs_mb.push_back(d->nonlinear_spring_m[i]); H.ncs_mb.push_back(d->nonlinear_spring_b[i]);
        for (int r = r0; r < r1; r++, row++) for (int k = 0; k < 7; k++) H.ncs_tab.push_back(d->nonlinear_spring_pieces[(size_t)7 * r + k]);
    }
    for (int i = 0; i < d->n_config_forces; i++) {
        H.cf_cfg.push_back(d->config_force_config[i]); H.cf_in.push_back(d->config_force_input[i]);
    }
    DevProg &P = H.p;
    P.nq = nq; P.nd = nd; P.nk = d->n_kin; P.nu = d->n_inputs; P.nc = nc; P.nf = nd + nc; P.nX = nq + nd + d->n_kin;
    P.n_joints = nj; P.n_levels = n_levels; P.n_bodies = nb; P.n_items = nitems; P.n_pairs = (int)H.pair_a.size();
    P.n_endpoints = (int)H.e_anchor.size(); P.n_dh = n_dh_con; P.n_dhr = (int)(H.dhr_pack.size() / 8); P.n_cf = (int)H.cf_cfg.size();
    P.n_springs = ns; P.n_sdh = n_dh_spr - n_dh_con;
    P.n_wrenches = nw; P.n_wdh = (int)H.dh_c.size() - n_dh_spr;
    P.n_cfgitems = (int)H.cfg_items.size();
    P.n_npairs = (int)(H.pair4.size() / 4);
    P.n_tri = (int)(H.tri4.size() / 4); P.n_cpair = n_cpair_con; P.n_spair = n_cpair_spr - n_cpair_con; P.n_wpair = (int)(H.cpair4.size() / 4) - n_cpair_spr;
    P.has_cs = (d->n_config_springs > 0 || d->n_nonlinear_springs > 0) ? 1 : 0;
    P.n_ncs = d->n_nonlinear_springs;
    P.n_cpath = (int)H.cpath_items.size();
    P.grav[0] = P.grav[1] = P.grav[2] = 0.0;
    for (int i = 0; i < d->n_gravity; i++)
        for (int k = 0; k < 3; k++) P.grav[k] += d->gravity[3 * (size_t)i + k];
    // LDS layout.  The Newton matrix is only alive between its assembly and the solve, the joint /
    // body poses only between the pose sweep and the Jacobians / constraints, so they share storage.
    int off = 0;
    auto take = [&](int n) { int o = off; off += (n > 0 ? n : 0); return o; };
    P.dh_ld = 0;
    P.o_q1 = take(nq); P.o_q2 = take(nq); P.o_p1 = take(nd); P.o_lam = take(nc); P.o_u = take(P.nu); P.o_dq = take(nq);
    P.o_f = take(P.nf);
    off = (off + 1) & ~1;     // J and W on an even double: 16-byte LDS accesses of an item's six doubles (mvi_core.hpp, ld6 / st6)
    P.o_J = take(6 * nitems); P.o_W = take(6 * nitems); P.o_vB = take(6 * nb); P.o_gam = take(3 * nb);
    P.o_Ldq = take(nd); P.o_Lddq = take(nd); P.o_Dh1 = take(P.n_dh); P.o_Dh2 = take(P.n_dh);  // compact: one value per (constraint, dependent config) item
    P.o_scal = take(P.nf); P.o_misc = take(2); P.o_nu = take(P.nu + P.nk);
    P.o_I = take(4 * nb);   // mass and principal inertias of every body (copied from the table once per kernel)
    P.o_ctol = take(nc);    // constraint tolerances, likewise
    P.o_sV = take(ns ? nq : 0); P.o_sH = take(P.n_spair);   // spring gradient per dynamic config, Hessian per item pair (midpoint)
    P.has_damper = d->n_linear_dampers > 0 ? 1 : 0; P.n_true_springs = n_true_springs;
    P.o_sX = take(P.has_damper ? P.n_sdh : 0); P.o_sVq = take(P.has_damper ? P.n_sdh : 0); P.o_sXX = take(P.has_damper ? P.n_spair : 0);
    P.o_svel = take(P.has_damper ? ns : 0); P.o_sF = take(P.has_damper ? nd : 0);
    P.o_wF = take(nw ? nd : 0); P.o_wH = take(2 * P.n_wpair); P.o_wD = take(6 * P.n_wdh);   // wrenches: generalized force, F_dq(a;b) and F_dq(b;a) per pair, (dp/dq, axis) per item
    // level schedule of the pose sweep: 16 packed words (own offset | parent offset << 16) per level, as ints
    P.n_chains = (int)H.ch_first.size(); P.n_rounds = (int)H.round_off.size() - 1;
    P.sched_ok = (12 * nj < 65536) ? 1 : 0;
    for (int r = 0; r < P.n_rounds; r++) if (H.round_off[r + 1] - H.round_off[r] > 16) P.sched_ok = 0;
    for (int c : H.ch_len) if (c >= 32768) P.sched_ok = 0;
    P.tab_ok = (nq < 4096 && nj < 4096 && 12 * (int)P.n_bodies <= 128 && 3 * (int)P.n_endpoints <= 64 && P.n_dh <= 128 && P.n_items <= 128 && P.nc <= 8 && P.nd + P.nc <= 64 &&
                6 * (int)P.n_bodies <= 64 && 2 * nj <= 128) ? 1 : 0;
    if (P.sched_ok) {   // 2: no round has more than 8 chains (the dual sweep gives each pose set half a wavefront)
        P.sched_ok = 2;
        for (int r = 0; r < P.n_rounds; r++) if (H.round_off[r + 1] - H.round_off[r] > 8) P.sched_ok = 1;
    }
    P.o_sched = take(P.sched_ok ? 16 * P.n_rounds : 0);   // two ints per (round, slot)
    std::vector<int> pose_need(nj, 0);      // bit 0: the midpoint pose of joint j is read (a body hangs below it), bit 1: its q2 pose (a constraint end point does)
    for (int k : H.it_joint) pose_need[k] |= 1;
    for (int e : H.e_anchor) for (int k = e; k >= 0; k = H.j_parent[k]) pose_need[k] |= 2;
    for (int k = nj - 1; k >= 0; k--) if (H.j_parent[k] >= 0) pose_need[H.j_parent[k]] |= pose_need[k];      // (ancestors: they are, by construction; kept explicit)
    for (int rot = 1; rot >= 0; rot--)
        for (int set = 0; set < 2; set++)
            for (int k = 0; k < nj; k++)
                if (((H.j_kind[k] >= TG_RX) ? 1 : 0) == rot && (pose_need[k] >> set & 1))
                    H.sj_list.push_back(H.j_cfg[k] | (H.j_kind[k] << 12) | (k << 16) | (set << 28));
    for (int rot = 1; rot >= 0; rot--)
        for (int set = 0; set < 2; set++)
            for (int k = 0; k < nj; k++)
                if (((H.j_kind[k] >= TG_RX) ? 1 : 0) == rot) H.sj_full.push_back(H.j_cfg[k] | (H.j_kind[k] << 12) | (k << 16) | (set << 28));
    if (H.sj_full.empty()) H.sj_full.push_back(0);
    P.n_sj = (int)H.sj_list.size();
    P.n_sj_rot = 0;
    for (int w : H.sj_list) P.n_sj_rot += (((w >> 12) & 0xF) >= TG_RX) ? 1 : 0;
    if (H.sj_list.empty()) H.sj_list.push_back(0);
    {   // quad-lane sweep plan: at most four rounds of at most four passes, at most 16 chain steps' worth of local-transform columns
        // in registers per round.  An instance is one chain of one pose set; only the needed ones (pose_need of the chain's first joint)
        P.sw_ok = (P.sched_ok && P.n_rounds >= 1 && P.n_rounds <= 4) ? 1 : 0;
        P.sw_maxlen = 0;
        for (int i = 0; i < 4; i++) P.sw_np[i] = 0;
        for (int i = 0; i < 16; i++) P.sw_len[i] = 0;
        for (int i = 0; i < 80; i++) P.sw_inst[i] = 0;
        for (int r = 0; r < P.n_rounds && P.sw_ok; r++) {
            const int nch = H.round_off[r + 1] - H.round_off[r];
            std::vector<int> inst;
            for (int slot = 0; slot < nch; slot++)
                for (int set = 0; set < 2; set++)
                    if (pose_need[H.ch_first[H.round_off[r] + slot]] >> set & 1) inst.push_back(slot | (set << 8) | (1 << 9));
            const int np = ((int)inst.size() + 4) / 5;
            if (np > 4 || nch > 15) { P.sw_ok = 0; break; }      // (slot 15 of a round stays empty: what a lane without an instance reads)
            P.sw_np[r] = np;
            int total = 0;
            for (int p = 0; p < np; p++) {
                int len = 0;
                for (int q = 0; q < 5 && 5 * p + q < (int)inst.size(); q++) {
                    P.sw_inst[(4 * r + p) * 5 + q] = inst[5 * p + q];
                    len = std::max(len, H.ch_len[H.round_off[r] + (inst[5 * p + q] & 0xFF)]);
                }
                P.sw_len[4 * r + p] = len; total += len;
                P.sw_maxlen = std::max(P.sw_maxlen, len);
            }
            if (total > 16) P.sw_ok = 0;
        }
    }
    P.df_ld = (P.nf + 1) | 1;  // augmented with the right-hand side; odd stride avoids LDS bank conflicts
    const int shared0 = off;
    P.o_Df = take(P.nf * P.df_ld);
    const int end_df = off;
    off = shared0;
    P.o_sc = take(2 * nj);
    P.o_G = take(std::max(12 * nj, 2 * nitems));  // also holds the per-item residual terms
    P.o_gB = take(12 * nb); P.o_pE = take(3 * P.n_endpoints);
    P.has_plane = has_plane ? 1 : 0; P.o_nE = take(has_plane ? 3 * nc : 0);   // world normals of the plane constraints
    P.o_wR = take(9 * d->n_hybrid_wrenches);   // world rotations of the wrench frames (body wrenches)
    P.o_dqi = take(nitems);  // per-item rates: alive only while the poses are (Jacobians -> prefix sums)
    off = std::max(off, end_df);
    P.lds_per_team = (off + 1) & ~1;
    // first-derivative kernel (MODE_DERIV1): full-width constraint Jacobians, the augmented KKT matrix
    // [Df | one right-hand side per derivative variable] and the two D.D2L2 tables
    off = P.lds_per_team;
    P.d_nrhs = nq + nd + P.nu + P.nk;
    P.d_o_Dh1 = take(nc * nq); P.d_o_Dh2 = take(nc * nq);
    P.d_aug_ld = (P.nf + P.d_nrhs + nc) | 1;  // + nc unit columns used by the second-derivative adjoint
    P.d_o_AUG = take(P.nf * P.d_aug_ld);
    {   // chunks of bodies for the prefix-sum form of the third-order body terms; scratch = the dead pose union
        P.o_tps = P.o_sc;
        // from the start of the pose / Newton-matrix union to the end of the base region, plus the two full-width constraint Jacobians
        // that follow it (dead after the KKT matrix's constant blocks)
        const int cap = (P.lds_per_team - P.o_sc) + (P.d_o_AUG - P.d_o_Dh1);
        H.tchunk.clear();
        H.tchunk.push_back(0);
        int b = 0;
        bool fits = true;
        while (b < nb && fits) {
            int b1 = b, items = 0;
            while (b1 < nb && 18 * (items + (H.b_item_off[b1 + 1] - H.b_item_off[b1]) + (b1 - b + 1)) <= cap) {
                items += H.b_item_off[b1 + 1] - H.b_item_off[b1];
                b1++;
            }
            if (b1 == b) fits = false; else { H.tchunk.push_back(b1); b = b1; }
        }
        if (!fits) { H.tchunk.clear(); H.tchunk.push_back(0); }
        if (fits && H.tchunk.size() > 2) {
            // same number of chunks, but cut where the (item, item) pair counts balance (the pair loop runs in passes of 64 or 128
            // lanes per chunk: two chunks of 221 pairs are four passes of two waves, 287 + 155 are five); kept if every chunk fits
            const int n_c = (int)H.tchunk.size() - 1, total = H.b_pair_off[nb];
            std::vector<int> cut(1, 0);
            for (int ci = 1; ci < n_c; ci++) {
                const long long target = (long long)total * ci / n_c;
                int b1 = cut.back() + 1;
                while (b1 < nb - (n_c - ci) && std::llabs((long long)H.b_pair_off[b1 + 1] - target) < std::llabs((long long)H.b_pair_off[b1] - target)) b1++;
                cut.push_back(b1);
            }
            cut.push_back(nb);
            bool ok_cut = true;
            for (int ci = 0; ci < n_c; ci++) {
                const int items = H.b_item_off[cut[ci + 1]] - H.b_item_off[cut[ci]];
                if (cut[ci + 1] <= cut[ci] || 18 * (items + (cut[ci + 1] - cut[ci])) > cap) ok_cut = false;
            }
            if (ok_cut) H.tchunk = cut;
        }
        P.n_tchunk = (int)H.tchunk.size() - 1;
    }
    {   // two-part pair lists of the helper-wave kernels
        // order[] = the indices first..last-1 rearranged as [part 0 | part 1]; returns the size of part 0
        auto split = [](int first, int last, const std::function<long long(int)> &key, std::vector<int> &order) {
            std::map<long long, std::vector<int>> groups;
            for (int n = first; n < last; n++) groups[key(n)].push_back(n);
            std::vector<const std::vector<int> *> by_size;
            for (auto &g : groups) by_size.push_back(&g.second);
            std::stable_sort(by_size.begin(), by_size.end(), [](const std::vector<int> *a, const std::vector<int> *b) { return a->size() > b->size(); });
            std::vector<int> part[2];
            for (auto *g : by_size) {
                std::vector<int> &dst = part[part[0].size() <= part[1].size() ? 0 : 1];
                dst.insert(dst.end(), g->begin(), g->end());
            }
            for (int h = 0; h < 2; h++) { std::sort(part[h].begin(), part[h].end()); order.insert(order.end(), part[h].begin(), part[h].end()); }
            return (int)part[0].size();
        };
        auto cfg_key = [&](int n) {
            const int a = H.it_cfg[H.pair_a[n]], b = H.it_cfg[H.pair_b[n]];
            return (long long)std::min(a, b) * 65536 + std::max(a, b);
        };
        std::vector<int> order;
        P.wp_split = split(0, (int)H.pair_a.size(), cfg_key, order);
        for (int n : order) { H.wp_a.push_back(H.pair_a[n]); H.wp_b.push_back(H.pair_b[n]); }
        for (int ci = 0; ci < P.n_tchunk; ci++) {
            const int first = H.b_pair_off[H.tchunk[ci]], last = H.b_pair_off[H.tchunk[ci + 1]];
            order.clear();
            H.wt_split.push_back(first + split(first, last, cfg_key, order));
            for (int n : order) { H.wt_a.push_back(H.pair_a[n]); H.wt_b.push_back(H.pair_b[n]); }
        }
        order.clear();
        P.wc_split = split(0, n_cpair_con, [&](int n) {
            const int a = H.cpair4[4 * n + 3] & 0xFFFF, b = H.cpair4[4 * n + 3] >> 16;
            return (long long)std::min(a, b) * 65536 + std::max(a, b); }, order);
        for (int n : order) for (int m = 0; m < 4; m++) H.wcp4.push_back(H.cpair4[4 * n + m]);
    }
    P.o_cps = (!has_plane && 6 * P.n_cpath + 6 * nc <= (P.o_gam + 3 * nb) - P.o_J) ? P.o_J : -1;   // (plane constraints: generic path)
    //   // J, W, vB, gam are dead there (recomputed afterwards)   // prefix / suffix sums of the constraint paths + per-constraint sums
    {   // MODE_DYN_DERIV1 layout on top of the base region
        int goff = P.lds_per_team;
        auto gtake = [&](int n) { int o = goff; goff += (n > 0 ? n : 0); return o; };
        P.g_nrhs = 2 * nq + P.nk + P.nu;
        P.g_ld = (P.nf + P.g_nrhs) | 1;
        P.g_o_AUG = gtake(P.nf * P.g_ld);
        P.g_o_P = 0; P.g_o_X = gtake(6 * nitems); P.g_o_aF = gtake(6 * nb); P.g_o_x = gtake(P.nf + nq);
        P.g_lds_per_team = (goff + 1) & ~1;
        P.g_max_cu = 0;
        for (int c = 0; c < nc; c++) P.g_max_cu = std::max(P.g_max_cu, H.cu_off[c + 1] - H.cu_off[c]);
    }
    P.gjc_ok = (std::max(12 * nj, 2 * nitems) >= 4 * 32) ? 1 : 0;   // gj_cols scratch (128 doubles) lives in the dead pose area
    P.d_o_T12 = take(nq * nd); P.d_o_T22 = take(nq * nd);
    P.d_lds_per_team = (off + 1) & ~1;
    {   // compact slice of the first-derivative kernel (DevProg::a_*)
        const int hole0 = P.o_sc, base_end = P.d_o_Dh1;     // the pose union and what follows it up to the end of the base region
        P.a_o_T12 = hole0;
        P.a_o_AUG = (std::max(base_end, hole0 + nq * nd) + 1) & ~1;
        P.a_aug_ld = (P.nf + P.d_nrhs) | 1;
        P.a_o_T22 = (P.a_o_AUG + P.nf * P.a_aug_ld + 1) & ~1;
        P.a_lds_per_team = (P.a_o_T22 + nq * nd + 1) & ~1;
        const bool plain = !P.has_cs && ns == 0 && nw == 0 && !has_plane && !P.has_damper;
        // (the solver's scratch moves from the pose area to the Jacobian area: 200 doubles)
        P.a_ok = (plain && 2 * nc * nq <= nq * nd && 6 * nitems >= 200 && P.a_lds_per_team < P.d_lds_per_team) ? 1 : 0;
        if (!P.a_ok) { P.a_o_T12 = P.d_o_T12; P.a_o_AUG = P.d_o_AUG; P.a_aug_ld = P.d_aug_ld; P.a_o_T22 = P.d_o_T22; P.a_lds_per_team = P.d_lds_per_team; }
    }
    // second-derivative (z-contracted) kernel: starts where the two D.D2L2 tables of the deriv1 layout are (they are
    // dead once the adjoint's right-hand side is formed); H11 / H22 are symmetric and stored packed.  Puppet-40:
    // 79.8 KB per trajectory -> two wavefronts per CU.
    off = P.d_o_T12;
    P.e_o_H11 = take(nq * (nq + 1) / 2); P.e_o_H22 = take(nq * (nq + 1) / 2); P.e_o_H12 = take(nq * (nq | 1)); P.e_o_G1 = take(nq * nc);
    P.e_o_w = take(P.nf); P.e_o_zq = take(nd); P.e_o_zp = take(nd); P.e_o_vec = take(12 * nq); P.e_o_vec2 = P.e_o_vec;   // tangent products of four columns: [3][4][nq]
    P.e_o_sT = take(P.has_damper ? 2 * P.n_spair : 0); P.e_o_sWX = take(P.has_damper ? 2 * ns : 0); P.e_o_sWXq = take(P.has_damper ? P.n_sdh : 0);
    P.e_o_wT = take(P.n_wpair); P.e_o_Hu = take(nw ? nq * P.nu : 0);   // point forces: w-contracted F.d3p per pair, -dt/2 w.F_dudq [nq][nu]
    off = std::max(off, P.d_lds_per_team);
    P.e_lds_per_team = (off + 1) & ~1;
    P.max_cfg_items = 0;
    for (int c = 0; c < nd; c++) P.max_cfg_items = std::max(P.max_cfg_items, H.cfg_item_off[c + 1] - H.cfg_item_off[c]);
    {   // composite form of the Newton matrix (see DevProg::cmp_*)
        P.cmp_ok = 0; P.n_cgroups = 0; P.n_cmpairs = 0; P.o_cmp = P.o_csw = P.o_ccz = P.o_cmpt = 0;
        for (int i = 0; i < 32; i++) P.cmp_gmask[i] = 0;
        H.cmp_rep.assign(std::max(nd, 1), -1); H.cmp_grp.assign(std::max(nd, 1), 0);
        std::vector<std::vector<int>> below(nd);            // bodies below each dynamic config
        for (int it = 0; it < nitems; it++) {
            const int c = H.it_cfg[it];
            if (c < nd) { below[c].push_back(H.it_body[it]); if (H.cmp_rep[c] < 0) H.cmp_rep[c] = it; }
        }
        std::map<std::vector<int>, int> group_of;
        H.cmp_goff.assign(1, 0);
        bool all_rep = nd > 0;
        for (int c = 0; c < nd; c++) {
            if (below[c].empty()) { all_rep = false; continue; }
            auto f = group_of.find(below[c]);
            if (f == group_of.end()) {
                f = group_of.emplace(below[c], (int)group_of.size()).first;
                for (int b : below[c]) H.cmp_gbody.push_back(b);
                H.cmp_goff.push_back((int)H.cmp_gbody.size());
            }
            H.cmp_grp[c] = f->second;
        }
        std::vector<unsigned char> from_pairs((size_t)nd * nd, 0), from_cmp((size_t)nd * nd, 0);
        for (int n = 0; n < P.n_npairs; n++) from_pairs[(size_t)(H.pair4[4 * n + 2] & 0xFFFF) * nd + (H.pair4[4 * n + 2] >> 16)] = 1;
        // the diagonal pairs (c, c) first, in config order: lane c of the first trip then holds config c's damping already (tdamp), and
        // every entry of the inertial block is written by exactly one lane -- plain stores, no LDS atomics
        for (int c = 0; c < nd && all_rep; c++) { H.cmp_pair.push_back(c | (c << 16)); from_cmp[(size_t)c * nd + c] = 1; }
        for (int c = 0; c < nd && all_rep; c++) {           // proper ancestors of c along the path of its representative item's body
            const int b = H.it_body[H.cmp_rep[c]];
            for (int it = H.b_item_off[b]; it < H.b_item_off[b + 1]; it++) {
                const int a = H.it_cfg[it];
                if (a == c) break;
                if (a < nd) { H.cmp_pair.push_back(a | (c << 16)); from_cmp[(size_t)a * nd + c] = 1; }
            }
        }
        const int need = 16 * (int)group_of.size() + 27 * nd;                      // J / W area
        if (all_rep && nd >= 8 && nd <= 64 && nb <= 32 && group_of.size() <= 32 && from_pairs == from_cmp && need <= 12 * nitems && nitems < 4096 &&
            17 * nb <= 2 * nj + std::max(12 * nj, 2 * nitems)) {      // (the per-body entries, 17 doubles apart: sin / cos + joint-pose area, dead by then)
            P.cmp_ok = 1; P.n_cgroups = (int)group_of.size(); P.n_cmpairs = (int)H.cmp_pair.size();
            P.o_cmp = P.o_J; P.o_csw = P.o_cmp + 16 * P.n_cgroups; P.o_ccz = P.o_csw + 12 * nd;
            for (auto &g : group_of) for (int b : g.first) P.cmp_gmask[g.second] |= 1 << b;
        }
    }
    {   // world-frame evaluation (DevProg::wev_*): needs the composite form (groups, pairs), every (body, path config) item on a
        // DYNAMIC config (a kinematic config above a body would need a twist and a rate but has no residual row), one lane per config
        // and per (body, axis), lists of at most 12 entries, and the twists + the zero record inside the J area (the q2 poses of the
        // dual sweep sit in the W area while they are read)
        P.wev_ok = 0; P.wev_depth = 0; P.wev_rc_ident = 1;
        for (int b = 0; b < nb; b++) {
            const double *C = &H.b_C[12 * (size_t)b];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) if (C[4 * i + j] != (i == j ? 1.0 : 0.0)) P.wev_rc_ident = 0;
        }
        H.wev_lane.assign(256, 0);
        bool ok = P.cmp_ok && P.tab_ok && P.sw_ok && P.sched_ok && nd + 3 * nb < 64 && nc > 0 && 6 * nitems >= 12 * nj && ns == 0 && nw == 0 && !P.has_cs;
        for (int it = 0; it < nitems && ok; it++) if (H.it_cfg[it] >= nd) ok = false;
        if (ok && 16 * P.n_cgroups + 12 * (nd + 1) > 6 * nitems) ok = false;
        if (ok && 12 * nj + 17 * nb > 6 * nitems + 9 * nb) ok = false;      // per-body world entries behind the q2 poses (W, vB, gam areas)
        if (ok && 16 * P.n_cgroups + 12 * (nd + 1) + 15 * nd > 12 * nitems) ok = false;
        std::vector<std::vector<int>> lists(64);
        if (ok) {
            for (int c = 0; c < nd; c++) {
                const int b = H.it_body[H.cmp_rep[c]];
                for (int it = H.b_item_off[b]; it < H.b_item_off[b + 1] && H.it_cfg[it] != c; it++) lists[c].push_back(H.it_cfg[it]);
                // a tree: the ancestors of config c are the same along every body's path
                for (int b2 = 0; b2 < nb; b2++) {
                    std::vector<int> other;
                    bool has = false;
                    for (int it = H.b_item_off[b2]; it < H.b_item_off[b2 + 1]; it++) { if (H.it_cfg[it] == c) { has = true; break; } other.push_back(H.it_cfg[it]); }
                    if (has && other != lists[c]) ok = false;
                }
            }
            for (int b = 0; b < nb; b++)
                for (int r = 0; r < 3; r++)
                    for (int it = H.b_item_off[b]; it < H.b_item_off[b + 1]; it++) lists[nd + 3 * b + r].push_back(H.it_cfg[it]);
            for (auto &l : lists) P.wev_depth = std::max(P.wev_depth, (int)l.size());
            if (P.wev_depth > 12 || P.wev_depth < 1) ok = false;
        }
        if (ok) {
            for (int l = 0; l < 64; l++) {
                int w[3] = {0, 0, 0};
                for (int e = 0; e < 12; e++) w[e >> 2] |= (e < (int)lists[l].size() ? lists[l][e] : nd) << (8 * (e & 3));
                for (int i = 0; i < 3; i++) H.wev_lane[4 * l + i] = w[i];
                if (l < nd) {
                    const int j = joint_of_cfg[l];
                    H.wev_lane[4 * l + 3] = (12 * j) | (H.j_kind[j] << 16) | (H.cmp_grp[l] << 24);
                } else if (l < nd + 3 * nb) H.wev_lane[4 * l + 3] = ((l - nd) / 3) | (((l - nd) % 3) << 8);
            }
            P.wev_ok = 1;
            P.o_ccz = P.o_csw + 12 * (nd + 1);      // one more twist record: the all-zero one the padded list entries point at
        }
    }
    {   // structured Newton solve: structural pattern of [[Df11, -Dh1^T], [Dh2, 0]] (newton_matrix in mvi_core.hpp writes exactly these
        // entries) -> bordered-block-diagonal plan.  Full-wave systems of the size gj_panel serves; the solver's scratch (trailing
        // system + border solution) lives in the dead J / W area.
        const int nf = P.nf;
        H.bbd_tab.assign(128, 0);
        for (int i = 0; i < 16; i++) P.bbd_tvar[i] = 0;
        P.bbd_ok = 0; P.bbd_g = P.bbd_ng = P.bbd_nb = P.bbd_t = 0; P.o_bbd = 0;
        if (nf >= 17 && nf <= 31 && nd >= 2) {
            std::vector<unsigned char> pat((size_t)nf * nf, 0);
            auto mark = [&](int i, int j) { if (i < nf && j < nf) { pat[(size_t)i * nf + j] = 1; pat[(size_t)j * nf + i] = 1; } };
            for (int i = 0; i < nd; i++) mark(i, i);
            for (int n = 0; n < P.n_npairs; n++) mark(H.pair4[4 * n + 2] & 0xFFFF, H.pair4[4 * n + 2] >> 16);
            for (int n = 0; n < n_dh_con; n++) if (H.dh_cfg[n] < nd) mark(H.dh_cfg[n], nd + H.dh_c[n]);
            for (size_t n = (size_t)n_cpair_con; n < H.cpair4.size() / 4; n++) {   // two-point springs / dampers, point forces
                const int ka = H.cpair4[4 * n + 3] & 0xFFFF, kb = H.cpair4[4 * n + 3] >> 16;
                if (ka < nd && kb < nd) mark(ka, kb);
            }
            H.newton_pattern = pat;
            const BbdPlan plan = bbd_plan(nf, nd, pat);
            if (plan.ok && plan.t * (plan.t + 2) <= 12 * nitems) {
                P.bbd_ok = 1; P.bbd_g = plan.g; P.bbd_ng = plan.ng; P.bbd_nb = plan.nb; P.bbd_t = plan.t;
                H.bbd_tab.assign(plan.tab, plan.tab + 128);
                for (int i = 0; i < 16; i++) P.bbd_tvar[i] = plan.tvar[i];
                P.o_bbd = P.lds_per_team;      // behind the base region; the derivative layouts (which start there) do not keep it
                P.lds_per_team += 64;
            }
        }
        {   // packed image (BbdPacked): only with the world-frame evaluation (its Newton-matrix phases are the writer), if it fits the union
            P.bbd_pk_ok = 0; P.bbd_pk_nr = P.bbd_pk_nc2 = P.bbd_pk_tb = P.bbd_pk_tc2 = P.bbd_pk_xs = P.bbd_pk_size = P.bbd_pk_nones = 0;
            H.bbd_map.assign(1, -1); H.bbd_ones.assign(1, -1); H.wev_pairx.assign(1, 0); H.wev_dhx.assign(1, 0);
            if (P.bbd_ok && P.wev_ok) {
                BbdPlan plan;
                plan.ok = 1; plan.g = P.bbd_g; plan.ng = P.bbd_ng; plan.nb = P.bbd_nb; plan.t = P.bbd_t;
                for (int i = 0; i < 16; i++) plan.tvar[i] = P.bbd_tvar[i];
                for (int i = 0; i < 128; i++) plan.tab[i] = H.bbd_tab[i];
                const BbdPacked K = bbd_pack(plan, nf);
                bool ok = K.ok && K.size <= P.nf * P.df_ld && K.size < 1024 && nd < 64 && P.n_dh < 256 && (int)K.ones.size() <= 64 &&
                          P.bbd_t * (P.bbd_t + 2) <= 17 * nb;        // (the solve's scratch moves behind the q2 poses: the twists stay intact for a fallback)
                std::vector<int> pairx, dhx;
                for (int n = 0; n < P.n_cmpairs && ok; n++) {
                    const int a = H.cmp_pair[n] & 0xFFFF, b = H.cmp_pair[n] >> 16;
                    const int ab = K.map[(size_t)a * (nf + 1) + b], ba = K.map[(size_t)b * (nf + 1) + a];
                    if (ab < 0 || ba < 0) { ok = false; break; }
                    pairx.push_back((int)((unsigned)a | ((unsigned)b << 6) | ((unsigned)ab << 12) | ((unsigned)ba << 22)));
                }
                for (int n = 0; n < P.n_dhr && ok; n++) {
                    const int c = H.dhr_pack[8 * (size_t)n], k = H.dhr_pack[8 * (size_t)n + 1], item = H.dhr_pack[8 * (size_t)n + 7];
                    const int kc = K.map[(size_t)k * (nf + 1) + nd + c], ck = K.map[(size_t)(nd + c) * (nf + 1) + k];
                    if (kc < 0 || ck < 0) { ok = false; break; }
                    dhx.push_back(item | (kc << 8) | (ck << 18));
                }
                for (int i = 0; i < nf && ok; i++) if (K.map[(size_t)i * (nf + 1) + nf] < 0) ok = false;
                if (ok) {
                    P.bbd_pk_ok = 1; P.bbd_pk_nr = K.nr; P.bbd_pk_nc2 = K.nc2; P.bbd_pk_tb = K.tb; P.bbd_pk_tc2 = K.tc2; P.bbd_pk_xs = P.o_scal - P.o_Df; P.bbd_pk_size = K.size;      // (the solution vector: the dense solvers' scale vector, nf doubles, unused by these kernels)
                    P.bbd_pk_nones = K.ones[0] < 0 ? 0 : (int)K.ones.size();
                    H.bbd_map = K.map; H.bbd_ones = K.ones; H.wev_pairx = pairx.empty() ? std::vector<int>(1, 0) : pairx; H.wev_dhx = dhx.empty() ? std::vector<int>(1, 0) : dhx;
                }
            }
        }
        if (P.cmp_ok) {     // representative (item | body << 16) of every dynamic config, staged like the plan tables (rollout kernels only)
            P.o_cmpt = P.lds_per_team;
            P.lds_per_team += (nd + 1) / 2;
            P.lds_per_team = (P.lds_per_team + 1) & ~1;
        }
    }
    H.pack();
    return H;
}

}  // namespace tg

namespace tg {
namespace detail {
template <typename T>
inline void pool_append(std::vector<T> &pool, std::vector<size_t> &offs, const std::vector<T> &v) {
    offs.push_back(pool.size());
    pool.insert(pool.end(), v.begin(), v.end());
    while (pool.size() % 4) pool.push_back(T());
}
}  // namespace detail

#define TG_INT_TABLES(X)                                                                                     \
    X(level_off) X(lvl_joints) X(round_off) X(ch_first) X(ch_len) X(ch_parent) X(j_parent) X(j_kind) X(j_cfg) X(j_pre_ident) X(b_anchor) X(b_item_off) X(b_pair_off) X(it_body) \
    X(it_joint) X(it_cfg) X(pair_a) X(pair_b) X(cfg_item_off) X(cfg_items) X(e_anchor) X(c_type) X(c_e1) X(c_e2) \
    X(c_cfg) X(c_comp) X(dh_c) X(dh_cfg) X(dh_joint) X(dh_side) X(cf_cfg) X(cf_in) X(dh_lookup) X(cu_off) X(it_slot) X(pair4) \
    X(tri4) X(cpair4) X(it_pack) X(dh_pack) X(cpath_off) X(cpath_items) X(dh_pos) X(tchunk) X(tri_off) X(wr_in) X(wr_kind) X(ncs_i) \
    X(wp_a) X(wp_b) X(wt_a) X(wt_b) X(wt_split) X(wcp4) X(bbd_tab) X(cmp_rep) X(cmp_grp) X(cmp_goff) X(cmp_gbody) X(cmp_pair) X(dhr_pack) X(at_i) X(ae_i) X(sj_list) X(sj_full) X(wev_lane) X(bbd_map) X(bbd_ones) X(wev_pairx) X(wev_dhx)
#define TG_DBL_TABLES(X) X(j_pre) X(jcoef) X(j_prm) X(at_d) X(ae_d) X(b_C) X(b_inertia) X(e_off) X(c_dist) X(c_tol) X(damp) X(cs_k) X(cs_kq0) X(cs_c0) X(s_k) X(s_x0) X(c_nloc) X(wr_const) X(s_c) X(wr_Rloc) X(ncs_mb) X(ncs_tab)

inline void HostProgram::pack() {
    ipool.clear(); dpool.clear(); ioff.clear(); doff.clear();
#define X(name) detail::pool_append(ipool, ioff, name);
    TG_INT_TABLES(X)
#undef X
#define X(name) detail::pool_append(dpool, doff, name);
    TG_DBL_TABLES(X)
#undef X
    ipool.push_back(0); dpool.push_back(0.0);
}

inline void HostProgram::bind(DevProg &P, const int *I, const double *D) const {
    size_t k = 0;
#define X(name) P.name = I + ioff[k++];
    TG_INT_TABLES(X)
#undef X
    k = 0;
#define X(name) P.name = D + doff[k++];
    TG_DBL_TABLES(X)
#undef X
    P.tab_i = I; P.tab_d = D; P.n_tab_i = (int)ipool.size(); P.n_tab_d = (int)dpool.size();
}
}  // namespace tg
