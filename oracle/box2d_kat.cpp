// Known-answer test for the from-scratch Box2D-subset solver (madrl_amd/csrc/multiwalker_core.hpp).
// TEST INFRASTRUCTURE ONLY.
//
// Box2D (erincatto/box2d v2.3.x, the version behind pybox2d 2.3 that multi_walker.py imports) is not vendored in the
// reference and not installed here, so the solver cannot be checked against the library itself.  What Box2D does publish
// is the output of its HelloWorld program (Box2D v2.3.0 User Manual, chapter 2 "Hello Box2D"): a 2 m x 2 m box of
// density 1, friction 0.3 dropped from y = 4 onto static ground whose top is y = 0, gravity (0, -10), 60 steps of
// 1/60 s with 6 velocity and 2 position iterations.  The manual prints "x y angle" with two decimals:
//      0.00 4.00 0.00 / 0.00 3.99 0.00 / 0.00 3.98 0.00 / ... / 0.00 1.25 0.00 / 0.00 1.13 0.00 / 0.00 1.01 0.00
// This file replays that scene through the env's own world_step (same polygon collide / contact-solver / integrator code
// that runs on the GPU; only the step parameters are overridden): the ground is the static 100 m x 20 m box of the
// manual (fixture A, created first), the falling box fixture B.
#define MW_FPS 60.0f
#define MW_VEL_ITERS 6
#define MW_POS_ITERS 2
#include "../madrl_amd/csrc/multiwalker_core.hpp"

#include <cstring>

using namespace mw;

extern "C" int b2kat_falling_box(float *xya, int steps) {
    static Model M;
    static World Wd;
    static Scratch S;
    std::memset(&M, 0, sizeof(M)); std::memset(&Wd, 0, sizeof(Wd)); std::memset(&S, 0, sizeof(S));
    M.W = 0; M.NB = 2; M.NJ = 0; M.NT = 2; M.max_manifolds = MAXM; M.continuous = 1;
    const V2 p[4] = {v2(-1, -1), v2(1, -1), v2(1, 1), v2(-1, 1)};  // SetAsBox(1, 1), density 1, friction 0.3
    poly_set(M.shape[SH_PACKAGE], p, 4);
    poly_mass(M.shape[SH_PACKAGE], 1.0f);
    M.shape[SH_PACKAGE].friction = 0.3f; M.shape[SH_PACKAGE].category = 0x001; M.shape[SH_PACKAGE].mask = 0xFFFF;
    const V2 g[4] = {v2(-50, -10), v2(50, -10), v2(50, 10), v2(-50, 10)};  // ground SetAsBox(50, 10), density 0: static
    Shape &gs = M.shape[SH_HULL];  // body 1 uses this shape slot
    poly_set(gs, g, 4);
    gs.centroid = v2(0, 0); gs.inv_mass = 0.0f; gs.inv_I = 0.0f; gs.friction = 0.2f; gs.category = 0x001; gs.mask = 0xFFFF;
    M.slot_base[0] = 0; M.slot_cap[0] = EDGE_SLOTS_PKG; M.slot_base[1] = EDGE_SLOTS_PKG; M.slot_cap[1] = EDGE_SLOTS_SMALL;
    M.dyn_slot_base = EDGE_SLOTS_PKG + EDGE_SLOTS_SMALL; M.n_dyn_pairs = 1;
    M.dyn_a[0] = 1; M.dyn_b[0] = 0;  // A = ground (the first proxy), B = box
    for (int k = 0; k < MAXSLOT; ++k) Wd.c.slot[k].edge = -1;
    Wd.c.ty[0] = Wd.c.ty[1] = -1000.0f;  // the terrain chain plays no part here
    const float x0 = 0.0f;
    Wd.h.b[0].c = v2(x0, 4.0f);
    Wd.h.b[1].c = v2(0.0f, -10.0f);
    SerialPar par;
    for (int i = 0; i < steps; ++i) {
        world_step(M, Wd.h, Wd.c, S, par);
        xya[3 * i] = Wd.h.b[0].c.x - x0; xya[3 * i + 1] = Wd.h.b[0].c.y; xya[3 * i + 2] = Wd.h.b[0].a;
    }
    return 0;
}
