// Host emulation of the device kernel source (TEST INFRASTRUCTURE ONLY).
//
// Compiles trep_amd/csrc/mvi_core.hpp with g++ and TEAM = 1 so that the CPU-only test-suite can
// check the kernel's arithmetic and control flow (everything except cross-lane races) against the
// oracle in a container without a GPU.  It is NOT part of the product: nothing in trep_amd/ loads it.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "../../trep_amd/csrc/mvi_core.hpp"

namespace {
struct Emu {
    tg::HostProgram H;
    tg::DevProg P;
};
}  // namespace

extern "C" {

void *emu_create(const tg_system_desc *d) {
    Emu *e = new Emu();
    try {
        e->H = tg::build_program(d);
    } catch (...) {
        delete e;
        return nullptr;
    }
    tg::HostProgram &H = e->H;
    e->P = H.p;
    tg::DevProg &P = e->P;
    P.level_off = H.level_off.data(); P.j_parent = H.j_parent.data(); P.j_kind = H.j_kind.data();
    P.j_cfg = H.j_cfg.data(); P.j_pre_ident = H.j_pre_ident.data(); P.j_pre = H.j_pre.data();
    P.b_anchor = H.b_anchor.data(); P.b_C = H.b_C.data(); P.b_inertia = H.b_inertia.data();
    P.b_item_off = H.b_item_off.data(); P.b_pair_off = H.b_pair_off.data();
    P.it_body = H.it_body.data(); P.it_joint = H.it_joint.data(); P.it_cfg = H.it_cfg.data();
    P.pair_a = H.pair_a.data(); P.pair_b = H.pair_b.data();
    P.cfg_item_off = H.cfg_item_off.data(); P.cfg_items = H.cfg_items.data();
    P.e_anchor = H.e_anchor.data(); P.e_off = H.e_off.data();
    P.c_type = H.c_type.data(); P.c_e1 = H.c_e1.data(); P.c_e2 = H.c_e2.data(); P.c_cfg = H.c_cfg.data();
    P.c_comp = H.c_comp.data(); P.c_dist = H.c_dist.data(); P.c_tol = H.c_tol.data();
    P.dh_lookup = H.dh_lookup.data(); P.cu_off = H.cu_off.data();
    P.dh_c = H.dh_c.data(); P.dh_cfg = H.dh_cfg.data(); P.dh_joint = H.dh_joint.data(); P.dh_side = H.dh_side.data();
    P.damp = H.damp.data(); P.cf_cfg = H.cf_cfg.data(); P.cf_in = H.cf_in.data();
    return e;
}

void emu_destroy(void *h) { delete (Emu *)h; }

int emu_lds_doubles(void *h) { return ((Emu *)h)->P.lds_per_team; }

// Runs every trajectory of the batch through the kernel body, one after the other.
void emu_run(void *h, const tg::RunArgs *args) {
    Emu *e = (Emu *)h;
    std::vector<double> lds((size_t)std::max(std::max(e->P.lds_per_team, e->P.d_lds_per_team), e->P.e_lds_per_team));
    for (int t = 0; t < args->batch; t++) {
        std::fill(lds.begin(), lds.end(), 0.0);
        tg::run_trajectory<1>(e->P, *args, lds.data(), 0, t);
    }
}
}
