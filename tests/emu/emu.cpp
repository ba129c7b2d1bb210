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

// The forward-mode kernels (run_forward: the continuous-dynamics modes on dual numbers), order 1 or 2.
template <class Real> static void emu_forward(Emu *e, const tg::RunArgs *args) {
    std::vector<Real> lds((size_t)std::max(e->P.lds_per_team, e->P.g_lds_per_team));
    for (int t = 0; t < args->batch; t++) {
        std::fill(lds.begin(), lds.end(), Real(0.0));
        switch (args->mode) {
        case tg::MODE_DYNAMICS: tg::run_forward<1, tg::MODE_DYNAMICS, true>(e->P, *args, lds.data(), 0, t); break;
        case tg::MODE_DYN_DERIV1: tg::run_forward<1, tg::MODE_DYN_DERIV1, true>(e->P, *args, lds.data(), 0, t); break;
        case tg::MODE_ENERGY: tg::run_forward<1, tg::MODE_ENERGY, true>(e->P, *args, lds.data(), 0, t); break;
        default: tg::run_forward<1, tg::MODE_LAGRANGIAN, true>(e->P, *args, lds.data(), 0, t); break;
        }
    }
}

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
    H.bind(P, H.ipool.data(), H.dpool.data());
    return e;
}

void emu_destroy(void *h) { delete (Emu *)h; }

int emu_lds_doubles(void *h) { return ((Emu *)h)->P.lds_per_team; }

// Runs every trajectory of the batch through the kernel body, one after the other.
void emu_run(void *h, const tg::RunArgs *args) {
    Emu *e = (Emu *)h;
    std::vector<double> lds((size_t)std::max(std::max(std::max(e->P.lds_per_team, e->P.d_lds_per_team), e->P.e_lds_per_team), e->P.g_lds_per_team));
    for (int t = 0; t < args->batch; t++) {
        std::fill(lds.begin(), lds.end(), 0.0);
        switch (args->mode) {
        case tg::MODE_ROLLOUT: tg::run_trajectory<1, tg::MODE_ROLLOUT, true>(e->P, *args, lds.data(), 0, t); break;
        case tg::MODE_CALC_P2: tg::run_trajectory<1, tg::MODE_CALC_P2, true>(e->P, *args, lds.data(), 0, t); break;
        case tg::MODE_CALC_F: tg::run_trajectory<1, tg::MODE_CALC_F, true>(e->P, *args, lds.data(), 0, t); break;
        case tg::MODE_DERIV1: tg::run_trajectory<1, tg::MODE_DERIV1, true>(e->P, *args, lds.data(), 0, t); break;
        case tg::MODE_DYNAMICS: tg::run_trajectory<1, tg::MODE_DYNAMICS, true>(e->P, *args, lds.data(), 0, t); break;
        case tg::MODE_DYN_DERIV1: tg::run_trajectory<1, tg::MODE_DYN_DERIV1, true>(e->P, *args, lds.data(), 0, t); break;
        case tg::MODE_ENERGY: tg::run_trajectory<1, tg::MODE_ENERGY, true>(e->P, *args, lds.data(), 0, t); break;
        case tg::MODE_LAGRANGIAN: tg::run_trajectory<1, tg::MODE_LAGRANGIAN, true>(e->P, *args, lds.data(), 0, t); break;
        default: tg::run_trajectory<1, tg::MODE_DERIV2Z, true>(e->P, *args, lds.data(), 0, t); break;
        }
    }
}

void emu_run_forward(void *h, const tg::RunArgs *args, int order) {
    if (order == 2) emu_forward<tgdual::Dual<tgdual::Dual<double>>>((Emu *)h, args);
    else emu_forward<tgdual::Dual<double>>((Emu *)h, args);
}
}
