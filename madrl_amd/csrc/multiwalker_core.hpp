// multiwalker_core.hpp -- MultiWalkerEnv dynamics: a from-scratch float32 restatement of the
// subset of Box2D 2.3.x that madrl_environments/walker/multi_walker.py drives
// (`self.world.Step(1.0 / FPS, 6 * 30, 2 * 30)`, multi_walker.py:365), plus the env logic
// around it.  Written once as host/device code: the HIP kernel (multiwalker.hip) runs it with the
// world resident in LDS.
//
// PARITY UNPINNED: the arithmetic lives in third-party Box2D (pybox2d /
// box2d-py, Box2D 2.3.x; no version pin anywhere in the reference tree, the module is not
// installable here and the reference has no golden vectors at that boundary).  What follows
// restates Box2D's published algorithms from their documented structure:
//   b2World::Step -> Collide -> Solve(islands) ; b2ContactSolver (sequential impulses, block
//   solver for 2-point manifolds, Baumgarte position correction) ; b2RevoluteJoint (point
//   constraint + motor + limit) ; b2CollidePolygons ; b2CollideEdgeAndPolygon ; b2EdgeShape::RayCast.
// Known, deliberate differences from Box2D (documented in DESIGN.md): no TOI / continuous pass,
// no sleeping, constraint order = this file's deterministic order (Box2D: creation/DFS order),
// lidar returns the closest terrain hit (Box2D reports the first hit in tree order).
//
// Reference call sites (file:line in /root/reference/madrl_environments/walker/multi_walker.py):
//   constants :17-47 ; BipedalWalker._reset :113-192 ; apply_action :194-203 ;
//   get_observation :205-237 ; ContactDetector :50-84 ; MultiWalkerEnv.setup/reset :276-357 ;
//   step :359-428 ; _generate_package :499-514 ; _generate_terrain :516-628.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MW_HD __host__ __device__ inline
#define MW_HD_INLINE __host__ __device__ __attribute__((always_inline)) inline  // must be inlined into the kernel: its World / Scratch references are LDS there, an out-of-line copy would fall back to flat addressing
#define MW_UNROLL _Pragma("unroll")
#else
#define MW_HD inline
#define MW_HD_INLINE inline
#define MW_UNROLL
#endif

namespace mw {

// optional counters of the CPU build (scripts/mw_stats.cpp): how many sub-slots / position iterations a step really runs
#ifdef MW_STATS
struct Stats { long steps, sub_a, sub_b, manifolds, merged, pos_iters, toi_full, toi_culled, toi_events, toi_undone, toi_vel_iters, toi_hist[10], toi_nisl[6]; };
extern Stats g_stats;
#define MW_STAT(f, v) (g_stats.f += (v))
#else
#define MW_STAT(f, v) ((void)0)
#endif

// World::Step parameters of the env (multi_walker.py:365).  Overridable ONLY by the known-answer harness of the test
// infrastructure, which replays the published Box2D HelloWorld scene (1/60 s, 6 velocity / 2 position iterations)
// through this same solver.
#ifndef MW_FPS
#define MW_FPS 50.0f
#define MW_VEL_ITERS (6 * 30)
#define MW_POS_ITERS (2 * 30)
#endif

// ---------------------------------------------------------------- env constants (:17-47)
constexpr float FPS = MW_FPS, SCALE = 30.0f;
constexpr float MOTORS_TORQUE = 80.0f, SPEED_HIP = 4.0f, SPEED_KNEE = 6.0f;
constexpr float LIDAR_RANGE = 160.0f / SCALE, INITIAL_RANDOM = 5.0f;
constexpr float LEG_DOWN = -8.0f / SCALE, LEG_W = 8.0f / SCALE, LEG_H = 34.0f / SCALE;
constexpr float PACKAGE_LENGTH = 240.0f;
constexpr float VIEWPORT_W = 600.0f, VIEWPORT_H = 400.0f;
constexpr float TERRAIN_STEP = 14.0f / SCALE;
constexpr int TERRAIN_LENGTH = 200, TERRAIN_GRASS = 10, TERRAIN_STARTPAD = 20;
constexpr float TERRAIN_HEIGHT = VIEWPORT_H / SCALE / 4.0f, FRICTION = 2.5f;
constexpr int WALKER_SEPERATION = 10;

// ---------------------------------------------------------------- Box2D constants (b2Settings.h)
constexpr float B2_PI = 3.14159265359f;
constexpr float LINEAR_SLOP = 0.005f, ANGULAR_SLOP = 2.0f / 180.0f * B2_PI, POLY_RADIUS = 2.0f * LINEAR_SLOP;
constexpr float MAX_LINEAR_CORRECTION = 0.2f, MAX_ANGULAR_CORRECTION = 8.0f / 180.0f * B2_PI;
constexpr float MAX_TRANSLATION = 2.0f, MAX_ROTATION = 0.5f * B2_PI, BAUMGARTE = 0.2f;
constexpr float GRAVITY_Y = -10.0f;  // b2World() default in pybox2d: gravity=(0,-10)
constexpr int VEL_ITERS = MW_VEL_ITERS, POS_ITERS = MW_POS_ITERS;

constexpr int MAX_WALKERS = 4;
constexpr int MAXB = 5 * MAX_WALKERS + 1;        // package + 5 bodies per walker
constexpr int MAXJ = 4 * MAX_WALKERS;
constexpr int MAXT = TERRAIN_LENGTH * MAX_WALKERS / 8;  // terrain points (:301)
constexpr int EDGE_SLOTS_SMALL = 6, EDGE_SLOTS_PKG = 36;
constexpr int MAXSLOT = (MAXB - 1) * EDGE_SLOTS_SMALL + EDGE_SLOTS_PKG + MAX_WALKERS * (MAX_WALKERS - 1) / 2 + MAX_WALKERS;
constexpr int MAXM = 36;  // largest active-manifold pool (Model::max_manifolds <= MAXM)

struct V2 { float x, y; };
MW_HD V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
MW_HD V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
MW_HD V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
MW_HD V2 operator-(V2 a) { return v2(-a.x, -a.y); }
MW_HD V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
MW_HD float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
MW_HD float cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
MW_HD V2 cross(V2 a, float s) { return v2(s * a.y, -s * a.x); }
MW_HD V2 cross(float s, V2 a) { return v2(-s * a.y, s * a.x); }
MW_HD float clampf(float a, float lo, float hi) { return fmaxf(lo, fminf(a, hi)); }
struct Rot { float s, c; };
// sin/cos from +,-,* only (Cody-Waite reduction by pi/2, cephes single-precision minimax
// polynomials on [-pi/4, pi/4]): the host build and the device build of this file then agree
// bit for bit, which libm's and the device library's sinf/cosf (each within an ulp or two of
// the true value, but not of each other) do not.
MW_HD void sincos_det(float a, float &sn, float &cs) {
    const float kf = floorf(a * 0.636619772f + 0.5f);  // nearest multiple of pi/2
    const int k = (int)kf;
    float r = (a - kf * 1.5703125f) - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.549789948768648e-8f;
    const float z = r * r;
    const float ps = r + r * z * ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f);
    const float pc = (1.0f - 0.5f * z) + z * z * ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f);
    switch (k & 3) {
        case 0: sn = ps; cs = pc; break;
        case 1: sn = pc; cs = -ps; break;
        case 2: sn = -ps; cs = -pc; break;
        default: sn = -pc; cs = ps; break;
    }
}
MW_HD Rot rot(float a) { Rot q; sincos_det(a, q.s, q.c); return q; }
MW_HD V2 mul(Rot q, V2 v) { return v2(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
MW_HD V2 mulT(Rot q, V2 v) { return v2(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
struct Xf { V2 p; Rot q; };
MW_HD V2 mul(Xf t, V2 v) { return mul(t.q, v) + t.p; }
MW_HD V2 mulT(Xf t, V2 v) { return mulT(t.q, v - t.p); }
MW_HD Rot mulT(Rot q, Rot r) { Rot o; o.s = q.c * r.s - q.s * r.c; o.c = q.c * r.c + q.s * r.s; return o; }
MW_HD Xf mulT(Xf A, Xf B) { Xf C; C.q = mulT(A.q, B.q); C.p = mulT(A.q, B.p - A.p); return C; }

}  // namespace mw
#include "multiwalker_toi.hpp"
namespace mw {

// ---------------------------------------------------------------- static model (per n_walkers)
enum { SH_PACKAGE = 0, SH_HULL = 1, SH_UPPER = 2, SH_LOWER = 3, N_SHAPES = 4 };
struct Shape {
    int n;
    V2 v[5], nrm[5];
    V2 centroid;      // = body localCenter (one fixture per body)
    float inv_mass, inv_I, friction;
    uint16_t category, mask;
};
struct JointDef {  // revoluteJointDef, multi_walker.py:145-179
    int bA, bB;
    V2 lA, lB;        // local anchors
    float lower, upper;
};
struct Model {
    int W, NB, NJ, NT;  // walkers, bodies, joints, terrain points
    int max_manifolds;  // size of the active-manifold pool of a step (Scratch::m)
    int continuous;     // b2World continuousPhysics (Box2D's default: on)
    Shape shape[N_SHAPES];
    JointDef jd[MAXJ];
    float package_length, package_scale;
    float start_x[MAX_WALKERS];
    int slot_base[MAXB], slot_cap[MAXB];  // body-vs-terrain manifold cache ranges
    int dyn_slot_base, n_dyn_pairs;
    int dyn_a[MAX_WALKERS * (MAX_WALKERS - 1) / 2 + MAX_WALKERS], dyn_b[MAX_WALKERS * (MAX_WALKERS - 1) / 2 + MAX_WALKERS];
};
MW_HD int shape_of_body(int b) { return b == 0 ? SH_PACKAGE : (((b - 1) % 5 == 0) ? SH_HULL : (((b - 1) % 5) % 2 == 1 ? SH_UPPER : SH_LOWER)); }
MW_HD int hull_of(int w) { return 1 + 5 * w; }
MW_HD int node_of(int b, int n_walkers) { return b == 0 ? n_walkers : (b - 1) / 5; }  // island graph node: walker index, or W for the package

// b2PolygonShape::Set (gift wrapping from the right-most, lowest vertex, CCW) + normals
inline void poly_set(Shape &s, const V2 *pts, int count) {
    int i0 = 0;
    float x0 = pts[0].x;
    for (int i = 1; i < count; ++i) {
        const float x = pts[i].x;
        if (x > x0 || (x == x0 && pts[i].y < pts[i0].y)) { i0 = i; x0 = x; }
    }
    int hull[8], m = 0, ih = i0;
    for (;;) {
        hull[m] = ih;
        int ie = 0;
        for (int j = 1; j < count; ++j) {
            if (ie == ih) { ie = j; continue; }
            const V2 r = pts[ie] - pts[hull[m]], v = pts[j] - pts[hull[m]];
            const float c = cross(r, v);
            if (c < 0.0f) ie = j;
            if (c == 0.0f && dot(v, v) > dot(r, r)) ie = j;
        }
        ++m;
        ih = ie;
        if (ie == i0) break;
    }
    s.n = m;
    for (int i = 0; i < m; ++i) s.v[i] = pts[hull[i]];
    for (int i = 0; i < m; ++i) {
        const V2 e = s.v[(i + 1) % m] - s.v[i];
        const float len = sqrtf(dot(e, e));
        s.nrm[i] = v2(e.y / len, -e.x / len);
    }
}
// b2PolygonShape::ComputeMass + b2Body::ResetMassData for a single-fixture body
inline void poly_mass(Shape &s, float density) {
    V2 center = v2(0, 0), ref = v2(0, 0);
    float area = 0.0f, I = 0.0f;
    for (int i = 0; i < s.n; ++i) ref = ref + s.v[i];
    ref = (1.0f / s.n) * ref;
    const float inv3 = 1.0f / 3.0f;
    for (int i = 0; i < s.n; ++i) {
        const V2 e1 = s.v[i] - ref, e2 = s.v[(i + 1) % s.n] - ref;
        const float D = cross(e1, e2), ta = 0.5f * D;
        area += ta;
        center = center + (ta * inv3) * (e1 + e2);
        const float intx2 = e1.x * e1.x + e2.x * e1.x + e2.x * e2.x, inty2 = e1.y * e1.y + e2.y * e1.y + e2.y * e2.y;
        I += (0.25f * inv3 * D) * (intx2 + inty2);
    }
    const float mass = density * area;
    center = (1.0f / area) * center;
    const V2 c = center + ref;
    float Io = density * I + mass * (dot(c, c) - dot(center, center));  // about the body origin
    Io -= mass * dot(c, c);                                              // about the centre of mass
    s.centroid = c;
    s.inv_mass = 1.0f / mass;
    s.inv_I = 1.0f / Io;
}

inline void build_model(Model &M, int n_walkers) {
    M.W = n_walkers; M.NB = 5 * n_walkers + 1; M.NJ = 4 * n_walkers;
    // observed maxima of simultaneously touching pairs over long random / collapsed rollouts: 18, 25, 34 for 2, 3, 4 walkers
    M.continuous = 1;
    M.max_manifolds = n_walkers <= 1 ? 16 : (n_walkers == 2 ? 24 : (n_walkers == 3 ? 28 : MAXM));
    M.NT = (int)(TERRAIN_LENGTH * n_walkers * 1 / 8.0);          // :301
    M.package_scale = n_walkers / 1.75f;                          // :293
    M.package_length = PACKAGE_LENGTH / SCALE * M.package_scale;  // :294
    const float init_x = TERRAIN_STEP * TERRAIN_STARTPAD / 2;
    for (int w = 0; w < n_walkers; ++w) M.start_x[w] = init_x + WALKER_SEPERATION * w * TERRAIN_STEP;  // :285-287
    {   // package :499-514
        const float hx = 120.0f * M.package_scale / SCALE, hy = 5.0f / SCALE;
        const V2 p[4] = {v2(-hx, hy), v2(hx, hy), v2(hx, -hy), v2(-hx, -hy)};
        poly_set(M.shape[SH_PACKAGE], p, 4);
        poly_mass(M.shape[SH_PACKAGE], 1.0f);
        M.shape[SH_PACKAGE].friction = 0.5f; M.shape[SH_PACKAGE].category = 0x004; M.shape[SH_PACKAGE].mask = 0xFFFF;
    }
    {   // hull :118-127
        const V2 p[5] = {v2(-30 / SCALE, 9 / SCALE), v2(6 / SCALE, 9 / SCALE), v2(34 / SCALE, 1 / SCALE), v2(34 / SCALE, -8 / SCALE), v2(-30 / SCALE, -8 / SCALE)};
        poly_set(M.shape[SH_HULL], p, 5);
        poly_mass(M.shape[SH_HULL], 5.0f);
        M.shape[SH_HULL].friction = 0.1f; M.shape[SH_HULL].category = 0x002; M.shape[SH_HULL].mask = 0xFFFF;
    }
    for (int k = 0; k < 2; ++k) {  // legs :136-163: SetAsBox order, default friction 0.2
        Shape &s = M.shape[k == 0 ? SH_UPPER : SH_LOWER];
        const float hx = (k == 0 ? 1.0f : 0.8f) * LEG_W / 2, hy = LEG_H / 2;
        s.n = 4;
        s.v[0] = v2(-hx, -hy); s.v[1] = v2(hx, -hy); s.v[2] = v2(hx, hy); s.v[3] = v2(-hx, hy);
        s.nrm[0] = v2(0, -1); s.nrm[1] = v2(1, 0); s.nrm[2] = v2(0, 1); s.nrm[3] = v2(-1, 0);
        poly_mass(s, 1.0f);
        s.friction = 0.2f; s.category = k == 0 ? 0x002 : 0x0020; s.mask = 0x001;
    }
    for (int w = 0; w < n_walkers; ++w)
        for (int side = 0; side < 2; ++side) {
            JointDef &hip = M.jd[4 * w + 2 * side], &knee = M.jd[4 * w + 2 * side + 1];
            hip.bA = hull_of(w); hip.bB = hull_of(w) + 1 + 2 * side;
            hip.lA = v2(0, LEG_DOWN); hip.lB = v2(0, LEG_H / 2); hip.lower = -0.8f; hip.upper = 1.1f;
            knee.bA = hip.bB; knee.bB = hip.bB + 1;
            knee.lA = v2(0, -LEG_H / 2); knee.lB = v2(0, LEG_H / 2); knee.lower = -1.6f; knee.upper = -0.1f;
        }
    int base = 0;
    for (int b = 0; b < M.NB; ++b) { M.slot_base[b] = base; M.slot_cap[b] = (b == 0) ? EDGE_SLOTS_PKG : EDGE_SLOTS_SMALL; base += M.slot_cap[b]; }
    M.dyn_slot_base = base;
    int np = 0;
    for (int w = 0; w < n_walkers; ++w) { M.dyn_a[np] = 0; M.dyn_b[np] = hull_of(w); ++np; }          // package (A) - hull
    for (int i = 0; i < n_walkers; ++i) for (int j = i + 1; j < n_walkers; ++j) { M.dyn_a[np] = hull_of(i); M.dyn_b[np] = hull_of(j); ++np; }
    M.n_dyn_pairs = np;
}

// ---------------------------------------------------------------- dynamic state
// ---------------------------------------------------------------- dynamic state
struct Body { V2 c; float a; V2 v; float w; };  // centre of mass, angle, velocities
struct Joint {
    float ix, iy, iz, motor_impulse;  // accumulated impulses (warm start)
    float motor_speed, max_torque;
    int limit_state;                  // 0 inactive, 1 at lower, 2 at upper, 3 equal
};
struct Slot {       // persistent manifold cache of one candidate pair (b2Contact); 32 bytes
    int16_t edge;   // terrain edge index, or -1: free slot (dyn pairs: always used)
    uint8_t npts, touching;
    uint32_t id[2];
    float ni[2], ti[2];
    // continuous pass (b2Contact e_toiFlag / e_enabledFlag / m_toiCount), valid within one step; m_toi lives in Cold::slot_toi
    uint8_t toi_flags;           // bit 0: the cached time of impact is valid, bit 1: disabled for the rest of this step
    uint8_t toi_count;
    uint16_t pad_;
};
struct Manifold {   // one active b2ContactVelocityConstraint + b2ContactPositionConstraint
    int8_t bA, bB;         // bA = -1: static terrain
    int16_t slot;
    uint8_t npts, type, block, pad_;  // type 0: faceA, 1: faceB; block: the 2-point block solver applies
    V2 local_normal, local_point, lp[2];  // b2Manifold (lp in the other body's frame)
    V2 normal, rA[2], rB[2];
    float friction, nm[2], tm[2], ni[2], ti[2];
    float k11, k12, k22, im11, im12, im22;  // block solver K and K^-1
};
static_assert(sizeof(Slot) == 32, "HBM record layout");
static_assert(sizeof(Manifold) == 140, "LDS budget of the HIP kernel: 4 envs x (Hot + Scratch) x 8 wavefronts per CU");
// The env state is split by how often a step touches it.
//   Hot:  bodies and flags -- read and written by every one of the 180 + 60 solver sweeps; the HIP kernel keeps it in
//         LDS for the duration of the step.
//   Cold: the joints' persistent state, the manifold cache (warm-start impulses of the candidate pairs) and the terrain
//         heights -- touched once per step (joint init / write-back, Collide, StoreImpulses, lidar); the HIP kernel
//         reads and writes it in place in HBM (L2).
struct Hot {
    Body b[MAXB];
    float push_x[MAX_WALKERS];    // ApplyForceToCenter pending until the first Step (:130-131)
    float prev_shaping[MAX_WALKERS], prev_package_shaping;
    uint8_t fallen[MAX_WALKERS], ground[MAX_WALKERS][2], game_over, pad_;
    uint32_t tick;
    int32_t t;
};
struct Cold {
    Joint j[MAXJ];                // warm-start impulses, motor targets, limit states: read at the start of a step into the
                                  // owning lane's JointCache, written back at its end
    Slot slot[MAXSLOT];
    float slot_toi[MAXSLOT];      // continuous pass: cached time of impact per slot (valid while Slot::toi_flags bit 0)
    V2 sweep_c0[MAXB];            // continuous pass: body centres / angles at the start of the step (b2Sweep::c0, a0) and the
    float sweep_a0[MAXB];         // time up to which a body has already been advanced (b2Sweep::alpha0)
    float sweep_alpha0[MAXB];
    float ty[MAXT];               // terrain heights (x = i * TERRAIN_STEP)
};
struct World { Hot h; Cold c; };  // the packed per-env record in HBM

constexpr int NDYN = MAX_WALKERS * (MAX_WALKERS - 1) / 2 + MAX_WALKERS;
constexpr int MAXSLOT_TERRAIN = (MAXB - 1) * EDGE_SLOTS_SMALL + EDGE_SLOTS_PKG;
struct Scratch {  // per-step workspace (LDS on the GPU)
    int nm;
    int8_t n_dyn, max_cnt, merge_ok, all_done;
    // mass data of the four shapes (package, hull, upper leg, lower leg), copied once per step
    float sh_im[N_SHAPES], sh_ii[N_SHAPES];
    V2 sh_lc[N_SHAPES];
    // solver schedule: terrain manifolds per body (indexed like the body's slots) and the active dynamic pairs
    uint8_t bm_cnt[MAXB], bm_idx[MAXSLOT_TERRAIN];
    int8_t dyn_midx[NDYN];   // manifold of pair p, or -1
    int8_t dyn_list[NDYN];   // the active pairs, in pair order ...
    int8_t dyn_owner[NDYN], dyn_man[NDYN], dyn_a_shape[NDYN];  // ... the body whose lane solves them (the pair's second body), their manifolds, the shape of the first body
    int8_t comp[MAX_WALKERS + 1];
    uint8_t isl_done[MAX_WALKERS + 1], walker_ok[MAX_WALKERS];
    float body_minsep[MAXB], dyn_minsep[NDYN];
    Manifold m[MAXM];  // LAST member: the HIP kernel allocates only Model::max_manifolds of them
};

MW_HD Xf body_xf(const Model &M, const Body &b, int bi) {
    Xf t; t.q = rot(b.a);
    t.p = b.c - mul(t.q, M.shape[shape_of_body(bi)].centroid);
    return t;
}
MW_HD Xf xf_from(V2 c, float a, V2 local_center) { Xf t; t.q = rot(a); t.p = c - mul(t.q, local_center); return t; }

// ---------------------------------------------------------------- narrow phase
struct ClipV { V2 v; uint32_t id; };
// contact feature id: indexA | indexB << 8 | typeA << 16 | typeB << 24  (type 0 vertex, 1 face)
MW_HD uint32_t mk_id(int ia, int ib, int ta, int tb) { return (uint32_t)ia | ((uint32_t)ib << 8) | ((uint32_t)ta << 16) | ((uint32_t)tb << 24); }
MW_HD uint32_t swap_id(uint32_t id) { return ((id >> 8) & 0xFF) | ((id & 0xFF) << 8) | (((id >> 24) & 0xFF) << 16) | (((id >> 16) & 0xFF) << 24); }

MW_HD int clip_segment(ClipV out[2], const ClipV in[2], V2 normal, float offset, int vertexIndexA) {
    int n = 0;
    const float d0 = dot(normal, in[0].v) - offset, d1 = dot(normal, in[1].v) - offset;
    if (d0 <= 0.0f) out[n++] = in[0];
    if (d1 <= 0.0f) out[n++] = in[1];
    if (d0 * d1 < 0.0f) {
        const float interp = d0 / (d0 - d1);
        out[n].v = in[0].v + interp * (in[1].v - in[0].v);
        out[n].id = mk_id(vertexIndexA, (in[0].id >> 8) & 0xFF, 0, 1);
        ++n;
    }
    return n;
}

struct ManifoldOut { int npts, type; V2 local_normal, local_point, lp[2]; uint32_t id[2]; };

MW_HD float find_max_separation(int *edge, const Shape &p1, Xf xf1, const Shape &p2, Xf xf2) {
    const Xf xf = mulT(xf2, xf1);
    int best = 0;
    float maxsep = -3.0e38f;
    for (int i = 0; i < p1.n; ++i) {
        const V2 n = mul(xf.q, p1.nrm[i]), v1 = mul(xf, p1.v[i]);
        float si = 3.0e38f;
        for (int j = 0; j < p2.n; ++j) { const float sij = dot(n, p2.v[j] - v1); if (sij < si) si = sij; }
        if (si > maxsep) { maxsep = si; best = i; }
    }
    *edge = best;
    return maxsep;
}

// b2CollidePolygons
MW_HD void collide_polygons(ManifoldOut &mo, const Shape &pA, Xf xfA, const Shape &pB, Xf xfB) {
    mo.npts = 0;
    const float total_radius = 2.0f * POLY_RADIUS;
    int edgeA = 0, edgeB = 0;
    const float sepA = find_max_separation(&edgeA, pA, xfA, pB, xfB);
    if (sepA > total_radius) return;
    const float sepB = find_max_separation(&edgeB, pB, xfB, pA, xfA);
    if (sepB > total_radius) return;
    const Shape *p1, *p2; Xf xf1, xf2; int edge1, flip;
    const float k_tol = 0.1f * LINEAR_SLOP;
    if (sepB > sepA + k_tol) { p1 = &pB; p2 = &pA; xf1 = xfB; xf2 = xfA; edge1 = edgeB; mo.type = 1; flip = 1; }
    else { p1 = &pA; p2 = &pB; xf1 = xfA; xf2 = xfB; edge1 = edgeA; mo.type = 0; flip = 0; }
    ClipV inc[2];
    {   // b2FindIncidentEdge
        const V2 normal1 = mulT(xf2.q, mul(xf1.q, p1->nrm[edge1]));
        int index = 0; float mind = 3.0e38f;
        for (int i = 0; i < p2->n; ++i) { const float dd = dot(normal1, p2->nrm[i]); if (dd < mind) { mind = dd; index = i; } }
        const int i1 = index, i2 = (i1 + 1 < p2->n) ? i1 + 1 : 0;
        inc[0].v = mul(xf2, p2->v[i1]); inc[0].id = mk_id(edge1, i1, 1, 0);
        inc[1].v = mul(xf2, p2->v[i2]); inc[1].id = mk_id(edge1, i2, 1, 0);
    }
    const int iv1 = edge1, iv2 = (edge1 + 1 < p1->n) ? edge1 + 1 : 0;
    V2 v11 = p1->v[iv1], v12 = p1->v[iv2];
    V2 local_tangent = v12 - v11;
    { const float len = sqrtf(dot(local_tangent, local_tangent)); local_tangent = (1.0f / len) * local_tangent; }
    const V2 local_normal = cross(local_tangent, 1.0f), plane_point = 0.5f * (v11 + v12);
    const V2 tangent = mul(xf1.q, local_tangent), normal = cross(tangent, 1.0f);
    v11 = mul(xf1, v11); v12 = mul(xf1, v12);
    const float front_offset = dot(normal, v11);
    const float side1 = -dot(tangent, v11) + total_radius, side2 = dot(tangent, v12) + total_radius;
    ClipV c1[2], c2[2];
    if (clip_segment(c1, inc, -tangent, side1, iv1) < 2) return;
    if (clip_segment(c2, c1, tangent, side2, iv2) < 2) return;
    mo.local_normal = local_normal; mo.local_point = plane_point;
    int n = 0;
    for (int i = 0; i < 2; ++i) {
        const float sep = dot(normal, c2[i].v) - front_offset;
        if (sep <= total_radius) {
            mo.lp[n] = mulT(xf2, c2[i].v);
            mo.id[n] = flip ? swap_id(c2[i].id) : c2[i].id;
            ++n;
        }
    }
    mo.npts = n;
}

// b2CollideEdgeAndPolygon (b2EPCollider::Collide); edge = shape A, whose body frame is the world.
// The terrain edges are given their neighbours' vertices (v0 before v1, v3 after v2) like the
// ghost vertices of a b2ChainShape.  The reference builds plain b2EdgeShapes; Box2D protects those
// from deep penetration with its continuous (TOI) pass, which this restatement does not have, and
// without it a foot pressed into a terrain vertex gets wedged by the internal-edge side normals.
// The adjacency test below is Box2D's own remedy for exactly that artefact.
MW_HD void collide_edge_polygon(ManifoldOut &mo, V2 v1, V2 v2e, const Shape &pB, Xf xfB, bool has0, V2 v0, bool has3, V2 v3) {
    mo.npts = 0;
    const Xf xf = xfB;  // edge body transform is identity
    const V2 centroidB = mul(xf, pB.centroid);
    V2 edge1 = v2e - v1;
    { const float len = sqrtf(dot(edge1, edge1)); edge1 = (1.0f / len) * edge1; }
    const V2 normal1 = v2(edge1.y, -edge1.x);
    const float offset1 = dot(normal1, centroidB - v1);
    float offset0 = 0.0f, offset2 = 0.0f;
    bool convex1 = false, convex2 = false;
    V2 normal0 = v2(0, 0), normal2 = v2(0, 0);
    if (has0) {
        V2 edge0 = v1 - v0;
        { const float len = sqrtf(dot(edge0, edge0)); edge0 = (1.0f / len) * edge0; }
        normal0 = v2(edge0.y, -edge0.x);
        convex1 = cross(edge0, edge1) >= 0.0f;
        offset0 = dot(normal0, centroidB - v0);
    }
    if (has3) {
        V2 edge2 = v3 - v2e;
        { const float len = sqrtf(dot(edge2, edge2)); edge2 = (1.0f / len) * edge2; }
        normal2 = v2(edge2.y, -edge2.x);
        convex2 = cross(edge1, edge2) > 0.0f;
        offset2 = dot(normal2, centroidB - v2e);
    }
    bool front;
    V2 m_normal, lower, upper;
    if (has0 && has3) {
        if (convex1 && convex2) {
            front = offset0 >= 0.0f || offset1 >= 0.0f || offset2 >= 0.0f;
            if (front) { m_normal = normal1; lower = normal0; upper = normal2; } else { m_normal = -normal1; lower = -normal1; upper = -normal1; }
        } else if (convex1) {
            front = offset0 >= 0.0f || (offset1 >= 0.0f && offset2 >= 0.0f);
            if (front) { m_normal = normal1; lower = normal0; upper = normal1; } else { m_normal = -normal1; lower = -normal2; upper = -normal1; }
        } else if (convex2) {
            front = offset2 >= 0.0f || (offset0 >= 0.0f && offset1 >= 0.0f);
            if (front) { m_normal = normal1; lower = normal1; upper = normal2; } else { m_normal = -normal1; lower = -normal1; upper = -normal0; }
        } else {
            front = offset0 >= 0.0f && offset1 >= 0.0f && offset2 >= 0.0f;
            if (front) { m_normal = normal1; lower = normal1; upper = normal1; } else { m_normal = -normal1; lower = -normal2; upper = -normal0; }
        }
    } else if (has0) {
        if (convex1) {
            front = offset0 >= 0.0f || offset1 >= 0.0f;
            if (front) { m_normal = normal1; lower = normal0; upper = -normal1; } else { m_normal = -normal1; lower = normal1; upper = -normal1; }
        } else {
            front = offset0 >= 0.0f && offset1 >= 0.0f;
            if (front) { m_normal = normal1; lower = normal1; upper = -normal1; } else { m_normal = -normal1; lower = normal1; upper = -normal0; }
        }
    } else if (has3) {
        if (convex2) {
            front = offset1 >= 0.0f || offset2 >= 0.0f;
            if (front) { m_normal = normal1; lower = -normal1; upper = normal2; } else { m_normal = -normal1; lower = -normal1; upper = normal1; }
        } else {
            front = offset1 >= 0.0f && offset2 >= 0.0f;
            if (front) { m_normal = normal1; lower = -normal1; upper = normal1; } else { m_normal = -normal1; lower = -normal2; upper = normal1; }
        }
    } else {
        front = offset1 >= 0.0f;
        if (front) { m_normal = normal1; lower = -normal1; upper = -normal1; } else { m_normal = -normal1; lower = normal1; upper = normal1; }
    }
    V2 bv[5], bn[5];
    for (int i = 0; i < pB.n; ++i) { bv[i] = mul(xf, pB.v[i]); bn[i] = mul(xf.q, pB.nrm[i]); }
    const float radius = 2.0f * POLY_RADIUS;
    // edge axis
    float edge_sep = 3.0e38f;
    for (int i = 0; i < pB.n; ++i) { const float s = dot(m_normal, bv[i] - v1); if (s < edge_sep) edge_sep = s; }
    if (edge_sep > radius) return;
    // polygon axis
    int poly_type = 0, poly_index = -1; float poly_sep = -3.0e38f;
    const V2 perp = v2(-m_normal.y, m_normal.x);
    for (int i = 0; i < pB.n; ++i) {
        const V2 n = -bn[i];
        const float s1 = dot(n, bv[i] - v1), s2 = dot(n, bv[i] - v2e), s = fminf(s1, s2);
        if (s > radius) { poly_type = 2; poly_index = i; poly_sep = s; break; }
        if (dot(n, perp) >= 0.0f) { if (dot(n - upper, m_normal) < -ANGULAR_SLOP) continue; }
        else { if (dot(n - lower, m_normal) < -ANGULAR_SLOP) continue; }
        if (s > poly_sep) { poly_type = 2; poly_index = i; poly_sep = s; }
    }
    if (poly_type != 0 && poly_sep > radius) return;
    const float k_rel = 0.98f, k_abs = 0.001f;
    const bool primary_edge = (poly_type == 0) || !(poly_sep > k_rel * edge_sep + k_abs);
    ClipV ie[2];
    int rf_i1, rf_i2; V2 rf_v1, rf_v2, rf_normal;
    if (primary_edge) {
        mo.type = 0;
        int best = 0; float bestv = dot(m_normal, bn[0]);
        for (int i = 1; i < pB.n; ++i) { const float val = dot(m_normal, bn[i]); if (val < bestv) { bestv = val; best = i; } }
        const int i1 = best, i2 = (i1 + 1 < pB.n) ? i1 + 1 : 0;
        ie[0].v = bv[i1]; ie[0].id = mk_id(0, i1, 1, 0);
        ie[1].v = bv[i2]; ie[1].id = mk_id(0, i2, 1, 0);
        if (front) { rf_i1 = 0; rf_i2 = 1; rf_v1 = v1; rf_v2 = v2e; rf_normal = normal1; }
        else { rf_i1 = 1; rf_i2 = 0; rf_v1 = v2e; rf_v2 = v1; rf_normal = -normal1; }
    } else {
        mo.type = 1;
        ie[0].v = v1; ie[0].id = mk_id(0, poly_index, 0, 1);
        ie[1].v = v2e; ie[1].id = mk_id(0, poly_index, 0, 1);
        rf_i1 = poly_index; rf_i2 = (rf_i1 + 1 < pB.n) ? rf_i1 + 1 : 0;
        rf_v1 = bv[rf_i1]; rf_v2 = bv[rf_i2]; rf_normal = bn[rf_i1];
    }
    const V2 side_n1 = v2(rf_normal.y, -rf_normal.x), side_n2 = -side_n1;
    const float so1 = dot(side_n1, rf_v1), so2 = dot(side_n2, rf_v2);
    ClipV c1[2], c2[2];
    if (clip_segment(c1, ie, side_n1, so1, rf_i1) < 2) return;
    if (clip_segment(c2, c1, side_n2, so2, rf_i2) < 2) return;
    if (primary_edge) { mo.local_normal = rf_normal; mo.local_point = rf_v1; }
    else { mo.local_normal = pB.nrm[rf_i1]; mo.local_point = pB.v[rf_i1]; }
    int n = 0;
    for (int i = 0; i < 2; ++i) {
        const float sep = dot(rf_normal, c2[i].v - rf_v1);
        if (sep <= radius) {
            if (primary_edge) { mo.lp[n] = mulT(xf, c2[i].v); mo.id[n] = c2[i].id; }
            else { mo.lp[n] = c2[i].v; mo.id[n] = swap_id(c2[i].id); }
            ++n;
        }
    }
    mo.npts = n;
}

MW_HD void body_aabb(const Model &M, const Hot &Wd, int bi, float &xmin, float &xmax, float &ymin, float &ymax) {
    const Shape &s = M.shape[shape_of_body(bi)];
    const Xf t = body_xf(M, Wd.b[bi], bi);
    xmin = ymin = 3.0e38f; xmax = ymax = -3.0e38f;
    for (int i = 0; i < s.n; ++i) {
        const V2 p = mul(t, s.v[i]);
        xmin = fminf(xmin, p.x); xmax = fmaxf(xmax, p.x); ymin = fminf(ymin, p.y); ymax = fmaxf(ymax, p.y);
    }
    xmin -= POLY_RADIUS; xmax += POLY_RADIUS; ymin -= POLY_RADIUS; ymax += POLY_RADIUS;
}

// ContactDetector.BeginContact / EndContact (:50-84) for one pair whose touching state changed
MW_HD void contact_event(const Model &M, Hot &Wd, int bA, int bB, bool begin) {
    // bA == -1: terrain
    for (int w = 0; w < M.W; ++w) {
        const int hull = hull_of(w);
        if (begin) {
            if (hull == bA && bB != 0) Wd.fallen[w] = 1;   // hull touches anything but the package
            if (hull == bB && bA != 0) Wd.fallen[w] = 1;
        }
        for (int k = 0; k < 2; ++k) {                      // legs[1], legs[3]: the lower legs
            const int leg = hull + 2 + 2 * k;
            if (leg == bA || leg == bB) Wd.ground[w][k] = begin ? 1 : 0;
        }
    }
    if (begin) {
        if (bA == 0 && !(bB >= 1 && (bB - 1) % 5 == 0)) Wd.game_over = 1;   // package touches a non-hull
        if (bB == 0 && !(bA >= 1 && (bA - 1) % 5 == 0)) Wd.game_over = 1;
    }
}

// ---------------------------------------------------------------- lane parallelism
// The step is written once for a group of cooperating lanes (`Par`): SerialPar (CPU build: one lane that owns
// everything, no-op sync) or a 16-lane group of a wavefront in the HIP kernel (four envs per wavefront).  Work is split
// so that lanes running concurrently never touch the same body -- bodies and their terrain contacts by lane, the two legs
// of a walker on two lanes, the package / hull contacts one at a time -- and every pair of constraints that shares a
// body keeps its serial Gauss-Seidel order, so the lane-parallel schedule produces bit-identical results to the serial
// one (constraints on disjoint bodies commute exactly).
// JOINTS = revolute joints a lane may own (joint ji belongs to lane ji % n()); their per-step constants and accumulated
// impulses live in lane-private storage (registers on the GPU) for the whole step.
struct SerialPar {
    static constexpr int JOINTS = MAXJ;   // joints a lane may own
    static constexpr int BODIES = MAXB;   // bodies a lane may own
    static constexpr bool MCACHE = false; // no lane-private manifold copy: the one lane owns every manifold
    MW_HD int lane() const { return 0; }
    MW_HD int n() const { return 1; }
    MW_HD void sync() const {}
    MW_HD int alloc(int *counter) const { return (*counter)++; }
};

// push an active manifold into the solver pool, carrying impulses over from the cached contact
template <class Par>
MW_HD int emit_manifold(const Model &M, Hot &Wd, Scratch &S, Par par, Slot &sl, int slot_index, const ManifoldOut &mo, int bA, int bB, float friction, int max_manifolds) {
    float ni[2] = {0, 0}, ti[2] = {0, 0};
    for (int i = 0; i < mo.npts; ++i)  // b2Contact::Update: match ids with the old manifold
        for (int k = 0; k < sl.npts; ++k)
            if (sl.id[k] == mo.id[i]) { ni[i] = sl.ni[k]; ti[i] = sl.ti[k]; break; }
    const bool touching = mo.npts > 0;
    if (touching != (sl.touching != 0)) contact_event(M, Wd, bA, bB, touching);
    sl.touching = touching; sl.npts = (uint8_t)mo.npts;
    for (int i = 0; i < mo.npts; ++i) { sl.id[i] = mo.id[i]; sl.ni[i] = ni[i]; sl.ti[i] = ti[i]; }
    if (!touching) return -1;
    const int idx = par.alloc(&S.nm);
    if (idx >= max_manifolds) return -1;  // pool exhausted: the pair is ignored this step
    Manifold &m = S.m[idx];
    m.bA = (int8_t)bA; m.bB = (int8_t)bB; m.slot = (int16_t)slot_index; m.npts = (uint8_t)mo.npts; m.type = (uint8_t)mo.type;
    m.local_normal = mo.local_normal; m.local_point = mo.local_point;
    for (int i = 0; i < mo.npts; ++i) { m.lp[i] = mo.lp[i]; m.ni[i] = ni[i]; m.ti[i] = ti[i]; }
    m.friction = friction;
    return idx;
}

// b2ContactManager::Collide for body `bi` against the terrain polyline (edge e spans x in [e, e+1] * TERRAIN_STEP).
// The body's cache is direct-mapped: edge e lives in slot e % cap (the candidate range is a run of consecutive edges no
// longer than the cache; a longer run loses its last edges for this step, like a full cache).
template <class Par>
MW_HD void collide_body_terrain(const Model &M, Hot &Wd, Cold &Cd, Scratch &S, Par par, int bi) {
    const Shape &s = M.shape[shape_of_body(bi)];
    float xmin, xmax, ymin, ymax;
    body_aabb(M, Wd, bi, xmin, xmax, ymin, ymax);
    int e0 = (int)floorf(xmin / TERRAIN_STEP), e1 = (int)floorf(xmax / TERRAIN_STEP);
    if (e0 < 0) e0 = 0;
    if (e1 > M.NT - 2) e1 = M.NT - 2;
    Slot *slots = Cd.slot + M.slot_base[bi];
    const int cap = M.slot_cap[bi];
    int cnt = 0;
    // contacts whose edge left the candidate range are destroyed (EndContact if touching)
    for (int k = 0; k < cap; ++k) {
        Slot &sl = slots[k];
        const int ed = sl.edge;
        if (ed >= 0 && (ed < e0 || ed > e1)) {
            if (sl.touching) contact_event(M, Wd, -1, bi, false);
            sl.edge = -1; sl.npts = 0; sl.touching = 0;
        }
    }
    const Xf xfB = body_xf(M, Wd.b[bi], bi);
    for (int e = e0; e <= e1; ++e) {
        const int k = e % cap;
        Slot &sl = slots[k];
        if (sl.edge >= 0 && sl.edge != e) continue;  // the run is longer than the cache: pair ignored this step
        if (sl.edge != e) { sl.edge = (int16_t)e; sl.npts = 0; sl.touching = 0; }
        const V2 p1 = v2(e * TERRAIN_STEP, Cd.ty[e]), p2 = v2((e + 1) * TERRAIN_STEP, Cd.ty[e + 1]);
        ManifoldOut mo; mo.npts = 0;
        const float elo = fminf(p1.y, p2.y) - POLY_RADIUS, ehi = fmaxf(p1.y, p2.y) + POLY_RADIUS;
        if (!(ymin > ehi + 0.2f || ymax < elo - 0.2f)) {
            const bool has0 = e > 0, has3 = e < M.NT - 2;
            const V2 p0 = has0 ? v2((e - 1) * TERRAIN_STEP, Cd.ty[e - 1]) : p1;
            const V2 p3 = has3 ? v2((e + 2) * TERRAIN_STEP, Cd.ty[e + 2]) : p2;
            collide_edge_polygon(mo, p1, p2, s, xfB, has0, p0, has3, p3);
        }
        const int idx = emit_manifold(M, Wd, S, par, sl, M.slot_base[bi] + k, mo, -1, bi, sqrtf(FRICTION * s.friction), M.max_manifolds);
        if (idx >= 0 && cnt < cap) S.bm_idx[M.slot_base[bi] + cnt++] = (uint8_t)idx;
    }
    S.bm_cnt[bi] = (uint8_t)cnt;
}

// package - hull and hull - hull pair p
template <class Par>
MW_HD void collide_dyn_pair(const Model &M, Hot &Wd, Cold &Cd, Scratch &S, Par par, int p) {
    const int bA = M.dyn_a[p], bB = M.dyn_b[p];
    const Shape &sA = M.shape[shape_of_body(bA)], &sB = M.shape[shape_of_body(bB)];
    Slot &sl = Cd.slot[M.dyn_slot_base + p];
    sl.edge = 0;
    float ax0, ax1, ay0, ay1, bx0, bx1, by0, by1;
    body_aabb(M, Wd, bA, ax0, ax1, ay0, ay1);
    body_aabb(M, Wd, bB, bx0, bx1, by0, by1);
    ManifoldOut mo; mo.npts = 0;
    if (!(ax0 > bx1 + 0.2f || bx0 > ax1 + 0.2f || ay0 > by1 + 0.2f || by0 > ay1 + 0.2f))
        collide_polygons(mo, sA, body_xf(M, Wd.b[bA], bA), sB, body_xf(M, Wd.b[bB], bB));
    S.dyn_midx[p] = (int8_t)emit_manifold(M, Wd, S, par, sl, M.dyn_slot_base + p, mo, bA, bB, sqrtf(sA.friction * sB.friction), M.max_manifolds);
}

// ---------------------------------------------------------------- island solver (b2Island::Solve)
// inverse mass, inverse inertia and local centre of the two bodies of a constraint; A = static terrain: zeros
struct MassAB { float mA, iA, mB, iB; V2 lcA, lcB; };
MW_HD MassAB mass_of_pair(const Scratch &S, int bA, int bB) {
    MassAB q;
    if (bA < 0) { q.mA = 0.0f; q.iA = 0.0f; q.lcA = v2(0, 0); }
    else { const int sa = shape_of_body(bA); q.mA = S.sh_im[sa]; q.iA = S.sh_ii[sa]; q.lcA = S.sh_lc[sa]; }
    const int sb = shape_of_body(bB);
    q.mB = S.sh_im[sb]; q.iB = S.sh_ii[sb]; q.lcB = S.sh_lc[sb];
    return q;
}

// k = [ex.x ex.y ex.z ey.x ey.y ey.z ez.x ez.y ez.z]; Cramer's rule as b2Mat33::Solve33, split into the part that only
// depends on the matrix (constant over the sweeps of a step) and the part that depends on the right-hand side
MW_HD void solve33_prepare(const float *k, float &cx, float &cy, float &cz, float &det) {
    const float exx = k[0], exy = k[1], exz = k[2], eyx = k[3], eyy = k[4], eyz = k[5], ezx = k[6], ezy = k[7], ezz = k[8];
    cx = eyy * ezz - eyz * ezy; cy = eyz * ezx - eyx * ezz; cz = eyx * ezy - eyy * ezx;  // cross(ey, ez)
    det = exx * cx + exy * cy + exz * cz;
    if (det != 0.0f) det = 1.0f / det;
}
MW_HD void solve33(const float *k, float cx, float cy, float cz, float det, float bx, float by, float bz, float &x, float &y, float &z) {
    const float exx = k[0], exy = k[1], exz = k[2], eyx = k[3], eyy = k[4], eyz = k[5], ezx = k[6], ezy = k[7], ezz = k[8];
    x = det * (bx * cx + by * cy + bz * cz);
    const float dx = by * ezz - bz * ezy, dy = bz * ezx - bx * ezz, dz = bx * ezy - by * ezx;        // cross(b, ez)
    y = det * (exx * dx + exy * dy + exz * dz);
    const float fx = eyy * bz - eyz * by, fy = eyz * bx - eyx * bz, fz = eyx * by - eyy * bx;        // cross(ey, b)
    z = det * (exx * fx + exy * fy + exz * fz);
}
MW_HD float solve22_prepare(const float *k) {
    const float a11 = k[0], a12 = k[3], a21 = k[1], a22 = k[4];
    float det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    return det;
}
MW_HD void solve22(const float *k, float det, float bx, float by, float &x, float &y) {
    const float a11 = k[0], a12 = k[3], a21 = k[1], a22 = k[4];
    x = det * (a22 * bx - a12 * by);
    y = det * (a11 * by - a21 * bx);
}

// b2ContactSolver::InitializeVelocityConstraints + WarmStart for manifold k
MW_HD void contact_init_warm(Hot &Wd, Manifold &m, const MassAB &q) {
    const float mA = q.mA, iA = q.iA, mB = q.mB, iB = q.iB;
    const V2 cA = m.bA < 0 ? v2(0, 0) : Wd.b[m.bA].c, cB = Wd.b[m.bB].c;
    Xf xfA; if (m.bA < 0) { xfA.p = v2(0, 0); xfA.q.s = 0; xfA.q.c = 1; } else xfA = xf_from(Wd.b[m.bA].c, Wd.b[m.bA].a, q.lcA);
    const Xf xfB = xf_from(Wd.b[m.bB].c, Wd.b[m.bB].a, q.lcB);
    V2 normal, pts[2];  // b2WorldManifold::Initialize
    if (m.type == 0) {
        normal = mul(xfA.q, m.local_normal);
        const V2 plane = mul(xfA, m.local_point);
        MW_UNROLL
        for (int i = 0; i < 2; ++i) if (i < m.npts) {
            const V2 clip = mul(xfB, m.lp[i]);
            const V2 a = clip + (POLY_RADIUS - dot(clip - plane, normal)) * normal, bb = clip - POLY_RADIUS * normal;
            pts[i] = 0.5f * (a + bb);
        }
    } else {
        normal = mul(xfB.q, m.local_normal);
        const V2 plane = mul(xfB, m.local_point);
        MW_UNROLL
        for (int i = 0; i < 2; ++i) if (i < m.npts) {
            const V2 clip = mul(xfA, m.lp[i]);
            const V2 bb = clip + (POLY_RADIUS - dot(clip - plane, normal)) * normal, a = clip - POLY_RADIUS * normal;
            pts[i] = 0.5f * (a + bb);
        }
        normal = -normal;
    }
    m.normal = normal;
    const V2 tangent = cross(normal, 1.0f);
    MW_UNROLL
        for (int i = 0; i < 2; ++i) if (i < m.npts) {
        m.rA[i] = pts[i] - cA; m.rB[i] = pts[i] - cB;
        const float rnA = cross(m.rA[i], normal), rnB = cross(m.rB[i], normal);
        const float kn = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
        m.nm[i] = kn > 0.0f ? 1.0f / kn : 0.0f;
        const float rtA = cross(m.rA[i], tangent), rtB = cross(m.rB[i], tangent);
        const float kt = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
        m.tm[i] = kt > 0.0f ? 1.0f / kt : 0.0f;
    }
    m.block = 0;
    if (m.npts == 2) {
        const float rn1A = cross(m.rA[0], normal), rn1B = cross(m.rB[0], normal), rn2A = cross(m.rA[1], normal), rn2B = cross(m.rB[1], normal);
        const float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B, k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
        const float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
        if (k11 * k11 < 1000.0f * (k11 * k22 - k12 * k12)) {
            m.k11 = k11; m.k12 = k12; m.k22 = k22;
            float det = k11 * k22 - k12 * k12;
            if (det != 0.0f) det = 1.0f / det;
            m.im11 = det * k22; m.im12 = -det * k12; m.im22 = det * k11;
            m.block = 1;
        } else {
            m.npts = 1;  // the constraints are redundant, just use one
        }
    }
    V2 vA = m.bA < 0 ? v2(0, 0) : Wd.b[m.bA].v, vB = Wd.b[m.bB].v;
    float wA = m.bA < 0 ? 0.0f : Wd.b[m.bA].w, wB = Wd.b[m.bB].w;
    MW_UNROLL
        for (int i = 0; i < 2; ++i) if (i < m.npts) {  // warm start
        const V2 P = m.ni[i] * normal + m.ti[i] * tangent;
        wA -= iA * cross(m.rA[i], P); vA = vA - mA * P;
        wB += iB * cross(m.rB[i], P); vB = vB + mB * P;
    }
    if (m.bA >= 0) { Wd.b[m.bA].v = vA; Wd.b[m.bA].w = wA; }
    Wd.b[m.bB].v = vB; Wd.b[m.bB].w = wB;
}

// One revolute joint for the duration of a step: the constants of b2RevoluteJoint::InitVelocityConstraints and the
// accumulated impulses, held by the lane that owns the joint (registers on the GPU) and written back once at the end.
struct JointCache {
    int bA, bB;
    float mA, iA, mB, iB;
    V2 lA, lB;                 // local anchors relative to the local centres
    float lower, upper;
    V2 rA, rB;
    float k[9], motor_mass;
    float c33x, c33y, c33z, idet33, idet22;  // the b-independent terms of b2Mat33::Solve33 / Solve22 on k
    float motor_speed, maxi;   // maxi = h * maxMotorTorque
    int limit_state;
    float ix, iy, iz, motor_impulse;
};

// b2RevoluteJoint::InitVelocityConstraints (+ warm start)
MW_HD void joint_init_warm(const Model &M, Hot &Wd, const Cold &Cd, Scratch &S, int ji, float h, JointCache &c) {
    const JointDef &jd = M.jd[ji];
    const Joint &j = Cd.j[ji];
    c.bA = jd.bA; c.bB = jd.bB;
    Body &A = Wd.b[jd.bA], &B = Wd.b[jd.bB];
    const MassAB q = mass_of_pair(S, jd.bA, jd.bB);
    const float mA = q.mA, iA = q.iA, mB = q.mB, iB = q.iB;
    c.mA = mA; c.iA = iA; c.mB = mB; c.iB = iB;
    c.lA = jd.lA - q.lcA; c.lB = jd.lB - q.lcB;
    c.lower = jd.lower; c.upper = jd.upper;
    const V2 rA = mul(rot(A.a), c.lA), rB = mul(rot(B.a), c.lB);
    c.rA = rA; c.rB = rB;
    float *k = c.k;
    k[0] = mA + mB + rA.y * rA.y * iA + rB.y * rB.y * iB;
    k[3] = -rA.y * rA.x * iA - rB.y * rB.x * iB;
    k[6] = -rA.y * iA - rB.y * iB;
    k[1] = k[3];
    k[4] = mA + mB + rA.x * rA.x * iA + rB.x * rB.x * iB;
    k[7] = rA.x * iA + rB.x * iB;
    k[2] = k[6]; k[5] = k[7];
    k[8] = iA + iB;
    float mm = iA + iB;
    if (mm > 0.0f) mm = 1.0f / mm;
    c.motor_mass = mm;
    solve33_prepare(k, c.c33x, c.c33y, c.c33z, c.idet33);
    c.idet22 = solve22_prepare(k);
    c.motor_speed = j.motor_speed; c.maxi = h * j.max_torque;
    c.ix = j.ix; c.iy = j.iy; c.iz = j.iz; c.motor_impulse = j.motor_impulse; c.limit_state = j.limit_state;
    const float angle = B.a - A.a;  // referenceAngle = 0 (the def is built from kwargs, not Initialize())
    if (fabsf(jd.upper - jd.lower) < 2.0f * ANGULAR_SLOP) c.limit_state = 3;
    else if (angle <= jd.lower) { if (c.limit_state != 1) c.iz = 0.0f; c.limit_state = 1; }
    else if (angle >= jd.upper) { if (c.limit_state != 2) c.iz = 0.0f; c.limit_state = 2; }
    else { c.limit_state = 0; c.iz = 0.0f; }
    const V2 P = v2(c.ix, c.iy);  // dtRatio = 1
    A.v = A.v - mA * P; A.w -= iA * (cross(rA, P) + c.motor_impulse + c.iz);
    B.v = B.v + mB * P; B.w += iB * (cross(rB, P) + c.motor_impulse + c.iz);
}

// b2RevoluteJoint::SolveVelocityConstraints
MW_HD void joint_solve_velocity(Hot &Wd, JointCache &c) {
    Body &A = Wd.b[c.bA], &B = Wd.b[c.bB];
    const float mA = c.mA, iA = c.iA, mB = c.mB, iB = c.iB;
    const V2 rA = c.rA, rB = c.rB;
    V2 vA = A.v, vB = B.v; float wA = A.w, wB = B.w;
    if (c.limit_state != 3) {  // motor (enableMotor is always true)
        const float Cdot = wB - wA - c.motor_speed;
        float imp = -c.motor_mass * Cdot;
        const float old = c.motor_impulse, maxi = c.maxi;
        c.motor_impulse = clampf(old + imp, -maxi, maxi);
        imp = c.motor_impulse - old;
        wA -= iA * imp; wB += iB * imp;
    }
    if (c.limit_state != 0) {  // limit + point constraint (3x3)
        const V2 Cdot1 = vB + cross(wB, rB) - vA - cross(wA, rA);
        const float Cdot2 = wB - wA;
        float ix, iy, iz;
        solve33(c.k, c.c33x, c.c33y, c.c33z, c.idet33, Cdot1.x, Cdot1.y, Cdot2, ix, iy, iz);
        ix = -ix; iy = -iy; iz = -iz;
        bool reduce = false;
        if (c.limit_state == 3) { c.ix += ix; c.iy += iy; c.iz += iz; }
        else if (c.limit_state == 1) { reduce = (c.iz + iz) < 0.0f; }
        else { reduce = (c.iz + iz) > 0.0f; }
        if (c.limit_state != 3) {
            if (reduce) {
                const float rx = -Cdot1.x + c.iz * c.k[6], ry = -Cdot1.y + c.iz * c.k[7];
                float qx, qy;
                solve22(c.k, c.idet22, rx, ry, qx, qy);
                ix = qx; iy = qy; iz = -c.iz;
                c.ix += qx; c.iy += qy; c.iz = 0.0f;
            } else { c.ix += ix; c.iy += iy; c.iz += iz; }
        }
        const V2 P = v2(ix, iy);
        vA = vA - mA * P; wA -= iA * (cross(rA, P) + iz);
        vB = vB + mB * P; wB += iB * (cross(rB, P) + iz);
    } else {  // point-to-point only
        const V2 Cdot = vB + cross(wB, rB) - vA - cross(wA, rA);
        float ix, iy;
        solve22(c.k, c.idet22, -Cdot.x, -Cdot.y, ix, iy);
        c.ix += ix; c.iy += iy;
        const V2 P = v2(ix, iy);
        vA = vA - mA * P; wA -= iA * cross(rA, P);
        vB = vB + mB * P; wB += iB * cross(rB, P);
    }
    A.v = vA; A.w = wA; B.v = vB; B.w = wB;
}

// b2ContactSolver::SolveVelocityConstraints for manifold k
// ... on velocities the caller holds (the continuous pass keeps its one moving body in registers over all sweeps)
MW_HD void contact_solve_velocity_on(Manifold &m, const MassAB &q, V2 &vA, float &wA, V2 &vB, float &wB) {
    const float mA = q.mA, iA = q.iA, mB = q.mB, iB = q.iB;
    const V2 normal = m.normal, tangent = cross(normal, 1.0f);
    MW_UNROLL
    for (int i = 0; i < 2; ++i) if (i < m.npts) {  // friction first
        const V2 dv = vB + cross(wB, m.rB[i]) - vA - cross(wA, m.rA[i]);
        const float vt = dot(dv, tangent);
        float lambda = m.tm[i] * (-vt);
        const float maxf = m.friction * m.ni[i];
        const float newi = clampf(m.ti[i] + lambda, -maxf, maxf);
        lambda = newi - m.ti[i];
        m.ti[i] = newi;
        const V2 P = lambda * tangent;
        vA = vA - mA * P; wA -= iA * cross(m.rA[i], P);
        vB = vB + mB * P; wB += iB * cross(m.rB[i], P);
    }
    if (m.npts == 1 || !m.block) {
        MW_UNROLL
    for (int i = 0; i < 2; ++i) if (i < m.npts) {
            const V2 dv = vB + cross(wB, m.rB[i]) - vA - cross(wA, m.rA[i]);
            const float vn = dot(dv, normal);
            float lambda = -m.nm[i] * (vn - 0.0f);  // restitution 0 -> velocityBias 0
            const float newi = fmaxf(m.ni[i] + lambda, 0.0f);
            lambda = newi - m.ni[i];
            m.ni[i] = newi;
            const V2 P = lambda * normal;
            vA = vA - mA * P; wA -= iA * cross(m.rA[i], P);
            vB = vB + mB * P; wB += iB * cross(m.rB[i], P);
        }
    } else {  // block solver
        const float a1 = m.ni[0], a2 = m.ni[1];
        const V2 dv1 = vB + cross(wB, m.rB[0]) - vA - cross(wA, m.rA[0]);
        const V2 dv2 = vB + cross(wB, m.rB[1]) - vA - cross(wA, m.rA[1]);
        float b1 = dot(dv1, normal), b2 = dot(dv2, normal);
        b1 -= m.k11 * a1 + m.k12 * a2;
        b2 -= m.k12 * a1 + m.k22 * a2;
        float x1 = 0, x2 = 0; bool ok = false;
        x1 = -(m.im11 * b1 + m.im12 * b2); x2 = -(m.im12 * b1 + m.im22 * b2);
        if (x1 >= 0.0f && x2 >= 0.0f) ok = true;
        if (!ok) { x1 = -m.nm[0] * b1; x2 = 0.0f; const float vn2 = m.k12 * x1 + b2; if (x1 >= 0.0f && vn2 >= 0.0f) ok = true; }
        if (!ok) { x1 = 0.0f; x2 = -m.nm[1] * b2; const float vn1 = m.k12 * x2 + b1; if (x2 >= 0.0f && vn1 >= 0.0f) ok = true; }
        if (!ok) { x1 = 0.0f; x2 = 0.0f; if (b1 >= 0.0f && b2 >= 0.0f) ok = true; }
        if (ok) {
            const float d1 = x1 - a1, d2 = x2 - a2;
            const V2 P1 = d1 * normal, P2 = d2 * normal;
            vA = vA - mA * (P1 + P2); wA -= iA * (cross(m.rA[0], P1) + cross(m.rA[1], P2));
            vB = vB + mB * (P1 + P2); wB += iB * (cross(m.rB[0], P1) + cross(m.rB[1], P2));
            m.ni[0] = x1; m.ni[1] = x2;
        }
    }
}
MW_HD void contact_solve_velocity(Hot &Wd, Manifold &m, const MassAB &q) {
    V2 vA = m.bA < 0 ? v2(0, 0) : Wd.b[m.bA].v, vB = Wd.b[m.bB].v;
    float wA = m.bA < 0 ? 0.0f : Wd.b[m.bA].w, wB = Wd.b[m.bB].w;
    contact_solve_velocity_on(m, q, vA, wA, vB, wB);
    if (m.bA >= 0) { Wd.b[m.bA].v = vA; Wd.b[m.bA].w = wA; }
    Wd.b[m.bB].v = vB; Wd.b[m.bB].w = wB;
}

// b2ContactSolver::SolvePositionConstraints for manifold k; returns its minimum separation
MW_HD float contact_solve_position(Hot &Wd, const Manifold &m, const MassAB &q) {
    float min_sep = 0.0f;
    const float mA = q.mA, iA = q.iA, mB = q.mB, iB = q.iB;
    V2 cA = m.bA < 0 ? v2(0, 0) : Wd.b[m.bA].c, cB = Wd.b[m.bB].c;
    float aA = m.bA < 0 ? 0.0f : Wd.b[m.bA].a, aB = Wd.b[m.bB].a;
    const V2 lcA = q.lcA, lcB = q.lcB;
    MW_UNROLL
    for (int i = 0; i < 2; ++i) if (i < m.npts) {
        const Xf xfA = xf_from(cA, aA, lcA), xfB = xf_from(cB, aB, lcB);
        V2 normal, point; float sep;
        if (m.type == 0) {
            normal = mul(xfA.q, m.local_normal);
            const V2 plane = mul(xfA, m.local_point), clip = mul(xfB, m.lp[i]);
            sep = dot(clip - plane, normal) - 2.0f * POLY_RADIUS; point = clip;
        } else {
            normal = mul(xfB.q, m.local_normal);
            const V2 plane = mul(xfB, m.local_point), clip = mul(xfA, m.lp[i]);
            sep = dot(clip - plane, normal) - 2.0f * POLY_RADIUS; point = clip;
            normal = -normal;
        }
        const V2 rA = point - cA, rB = point - cB;
        min_sep = fminf(min_sep, sep);
        const float C = clampf(BAUMGARTE * (sep + LINEAR_SLOP), -MAX_LINEAR_CORRECTION, 0.0f);
        const float rnA = cross(rA, normal), rnB = cross(rB, normal);
        const float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
        const float imp = K > 0.0f ? -C / K : 0.0f;
        const V2 P = imp * normal;
        cA = cA - mA * P; aA -= iA * cross(rA, P);
        cB = cB + mB * P; aB += iB * cross(rB, P);
    }
    if (m.bA >= 0) { Wd.b[m.bA].c = cA; Wd.b[m.bA].a = aA; }
    Wd.b[m.bB].c = cB; Wd.b[m.bB].a = aB;
    return min_sep;
}

// b2RevoluteJoint::SolvePositionConstraints; returns whether the joint is within tolerance
MW_HD bool joint_solve_position(Hot &Wd, const JointCache &c) {
    Body &A = Wd.b[c.bA], &B = Wd.b[c.bB];
    const float mA = c.mA, iA = c.iA, mB = c.mB, iB = c.iB;
    float ang_err = 0.0f;
    if (c.limit_state != 0) {
        const float angle = B.a - A.a;
        float limit_imp = 0.0f;
        if (c.limit_state == 3) {
            const float C = clampf(angle - c.lower, -MAX_ANGULAR_CORRECTION, MAX_ANGULAR_CORRECTION);
            limit_imp = -c.motor_mass * C; ang_err = fabsf(C);
        } else if (c.limit_state == 1) {
            float C = angle - c.lower; ang_err = -C;
            C = clampf(C + ANGULAR_SLOP, -MAX_ANGULAR_CORRECTION, 0.0f);
            limit_imp = -c.motor_mass * C;
        } else {
            float C = angle - c.upper; ang_err = C;
            C = clampf(C - ANGULAR_SLOP, 0.0f, MAX_ANGULAR_CORRECTION);
            limit_imp = -c.motor_mass * C;
        }
        A.a -= iA * limit_imp; B.a += iB * limit_imp;
    }
    const V2 rA = mul(rot(A.a), c.lA), rB = mul(rot(B.a), c.lB);
    const V2 C = B.c + rB - A.c - rA;
    const float pos_err = sqrtf(dot(C, C));
    const float kxx = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y, kxy = -iA * rA.x * rA.y - iB * rB.x * rB.y;
    const float kyy = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
    float det = kxx * kyy - kxy * kxy;
    if (det != 0.0f) det = 1.0f / det;
    const V2 imp = v2(-(det * (kyy * C.x - kxy * C.y)), -(det * (kxx * C.y - kxy * C.x)));
    A.c = A.c - mA * imp; A.a -= iA * cross(rA, imp);
    B.c = B.c + mB * imp; B.a += iB * cross(rB, imp);
    return pos_err <= LINEAR_SLOP && ang_err <= ANGULAR_SLOP;
}

// ---------------------------------------------------------------- continuous pass (b2World::SolveTOI)
// After the discrete solve Box2D looks, for every contact between a dynamic body and a static one (here: terrain edges; in
// the known-answer scene the static ground box), for the first time in this step at which the two shapes come within
// linearSlop of each other (multiwalker_toi.hpp), takes the earliest such event, moves the body back to that time, solves
// a sub-step for the remaining time on a mini island (that body and its touching static contacts: 20 TOI position
// iterations at Baumgarte 0.75, the step's velocity iterations without warm starting, integration) and repeats until no
// event is left (at most 8 sub-steps per contact).  Joints take no part (b2Island::SolveTOI ignores them).
MW_HD void poly_aabb_at(const Shape &s, V2 c, float a, float &xmin, float &xmax, float &ymin, float &ymax) {
    const Xf t = xf_from(c, a, s.centroid);
    for (int i = 0; i < s.n; ++i) {
        const V2 p = mul(t, s.v[i]);
        xmin = fminf(xmin, p.x); xmax = fmaxf(xmax, p.x); ymin = fminf(ymin, p.y); ymax = fmaxf(ymax, p.y);
    }
}
MW_HD void proxy_of_shape(Proxy &p, const Shape &s) {
    p.n = s.n;
    MW_UNROLL
    for (int i = 0; i < TOI_MAX_VERTS; ++i) p.v[i] = i < s.n ? s.v[i] : s.v[0];
}
MW_HD Sweep sweep_of_body(const Model &M, const Hot &Wd, const Cold &Cd, int b) {
    Sweep s;
    s.lc = M.shape[shape_of_body(b)].centroid;
    s.c0 = Cd.sweep_c0[b]; s.a0 = Cd.sweep_a0[b]; s.alpha0 = Cd.sweep_alpha0[b];
    s.c = Wd.b[b].c; s.a = Wd.b[b].a;
    return s;
}
// time of impact of body `bi` (dynamic) with terrain edge e, as b2World::SolveTOI computes it for one contact
MW_HD float toi_alpha_terrain(const Model &M, const Hot &Wd, const Cold &Cd, int bi, int e, const Sweep &sB, float xmin, float xmax, float ymin, float ymax) {
    const V2 p1 = v2(e * TERRAIN_STEP, Cd.ty[e]), p2 = v2((e + 1) * TERRAIN_STEP, Cd.ty[e + 1]);
    // conservative cull (never changes a result): the swept vertex box of the body, inflated by what TOI calls touching, misses the edge
    const float m = 4.0f * LINEAR_SLOP;
    if (xmin - m > p2.x || xmax + m < p1.x || ymin - m > fmaxf(p1.y, p2.y) || ymax + m < fminf(p1.y, p2.y)) { MW_STAT(toi_culled, 1); return 1.0f; }
    MW_STAT(toi_full, 1);
    Proxy pA, pB;
    pA.n = 2; pA.v[0] = p1; pA.v[1] = p2;
    MW_UNROLL
    for (int i = 2; i < TOI_MAX_VERTS; ++i) pA.v[i] = p1;
    proxy_of_shape(pB, M.shape[shape_of_body(bi)]);
    Sweep sA;
    sA.lc = v2(0, 0); sA.c0 = v2(0, 0); sA.c = v2(0, 0); sA.a0 = 0.0f; sA.a = 0.0f; sA.alpha0 = 0.0f;
    float beta;
    const int state = time_of_impact(beta, pA, sA, pB, sB);
    const float alpha0 = sB.alpha0;
    return state == TOI_TOUCHING ? fminf(alpha0 + (1.0f - alpha0) * beta, 1.0f) : 1.0f;
}
// the same for dynamic pair p when one of its bodies is static (known-answer scene)
MW_HD float toi_alpha_pair(const Model &M, const Hot &Wd, const Cold &Cd, int bA, int bB) {
    Proxy pA, pB;
    proxy_of_shape(pA, M.shape[shape_of_body(bA)]); proxy_of_shape(pB, M.shape[shape_of_body(bB)]);
    Sweep sA = sweep_of_body(M, Wd, Cd, bA), sB = sweep_of_body(M, Wd, Cd, bB);
    float alpha0 = sA.alpha0;
    if (sA.alpha0 < sB.alpha0) { alpha0 = sB.alpha0; sweep_advance(sA, alpha0); }
    else if (sB.alpha0 < sA.alpha0) { alpha0 = sA.alpha0; sweep_advance(sB, alpha0); }
    float beta;
    const int state = time_of_impact(beta, pA, sA, pB, sB);
    return state == TOI_TOUCHING ? fminf(alpha0 + (1.0f - alpha0) * beta, 1.0f) : 1.0f;
}
// b2Contact::Update of a cached pair at the bodies' current poses: new manifold, impulses carried over by feature id,
// Begin / EndContact; returns whether the pair touches.  bA < 0: terrain edge `sl.edge`.
MW_HD bool toi_update_contact(const Model &M, Hot &Wd, const Cold &Cd, Slot &sl, int bA, int bB, ManifoldOut &mo) {
    mo.npts = 0;
    const Shape &sB = M.shape[shape_of_body(bB)];
    const Xf xfB = body_xf(M, Wd.b[bB], bB);
    if (bA < 0) {
        const int e = sl.edge;
        const V2 p1 = v2(e * TERRAIN_STEP, Cd.ty[e]), p2 = v2((e + 1) * TERRAIN_STEP, Cd.ty[e + 1]);
        const bool has0 = e > 0, has3 = e < M.NT - 2;
        const V2 p0 = has0 ? v2((e - 1) * TERRAIN_STEP, Cd.ty[e - 1]) : p1;
        const V2 p3 = has3 ? v2((e + 2) * TERRAIN_STEP, Cd.ty[e + 2]) : p2;
        collide_edge_polygon(mo, p1, p2, sB, xfB, has0, p0, has3, p3);
    } else {
        collide_polygons(mo, M.shape[shape_of_body(bA)], body_xf(M, Wd.b[bA], bA), sB, xfB);
    }
    float ni[2] = {0, 0}, ti[2] = {0, 0};
    for (int i = 0; i < mo.npts; ++i)
        for (int k = 0; k < sl.npts; ++k)
            if (sl.id[k] == mo.id[i]) { ni[i] = sl.ni[k]; ti[i] = sl.ti[k]; break; }
    const bool touching = mo.npts > 0;
    if (touching != (sl.touching != 0)) contact_event(M, Wd, bA, bB, touching);
    sl.touching = touching; sl.npts = (uint8_t)mo.npts;
    for (int i = 0; i < mo.npts; ++i) { sl.id[i] = mo.id[i]; sl.ni[i] = ni[i]; sl.ti[i] = ti[i]; }
    return touching;
}
// b2ContactSolver::SolveTOIPositionConstraints for one manifold: only the TOI body (B; A is static) moves
MW_HD float contact_solve_toi_position(Hot &Wd, const Manifold &m, const MassAB &q) {
    float min_sep = 0.0f;
    const float mB = q.mB, iB = q.iB;
    const V2 cA = m.bA < 0 ? v2(0, 0) : Wd.b[m.bA].c;
    const float aA = m.bA < 0 ? 0.0f : Wd.b[m.bA].a;
    V2 cB = Wd.b[m.bB].c;
    float aB = Wd.b[m.bB].a;
    MW_UNROLL
    for (int i = 0; i < 2; ++i) if (i < m.npts) {
        const Xf xfA = xf_from(cA, aA, q.lcA), xfB = xf_from(cB, aB, q.lcB);
        V2 normal, point; float sep;
        if (m.type == 0) {
            normal = mul(xfA.q, m.local_normal);
            const V2 plane = mul(xfA, m.local_point), clip = mul(xfB, m.lp[i]);
            sep = dot(clip - plane, normal) - 2.0f * POLY_RADIUS; point = clip;
        } else {
            normal = mul(xfB.q, m.local_normal);
            const V2 plane = mul(xfB, m.local_point), clip = mul(xfA, m.lp[i]);
            sep = dot(clip - plane, normal) - 2.0f * POLY_RADIUS; point = clip;
            normal = -normal;
        }
        const V2 rB = point - cB;
        min_sep = fminf(min_sep, sep);
        const float C = clampf(0.75f * (sep + LINEAR_SLOP), -MAX_LINEAR_CORRECTION, 0.0f);  // b2_toiBaugarte
        const float rnB = cross(rB, normal);
        const float K = mB + iB * rnB * rnB;   // the static body contributes nothing
        const float imp = K > 0.0f ? -C / K : 0.0f;
        const V2 P = imp * normal;
        cB = cB + mB * P; aB += iB * cross(rB, P);
    }
    Wd.b[m.bB].c = cB; Wd.b[m.bB].a = aB;
    return min_sep;
}

constexpr int MAX_TOI_CONTACTS = 32;  // b2_maxTOIContacts
constexpr int MAX_SUB_STEPS = 8;      // b2_maxSubSteps

// (re)compute the invalidated times of impact of a body's terrain contacts and its earliest remaining event
MW_HD void toi_refresh_body(const Model &M, const Hot &Wd, Cold &Cd, Scratch &S, int bi) {
    const int base = M.slot_base[bi], cap = M.slot_cap[bi];
    const Sweep sB = sweep_of_body(M, Wd, Cd, bi);
    const Shape &sh = M.shape[shape_of_body(bi)];
    float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
    poly_aabb_at(sh, sB.c0, sB.a0, xmin, xmax, ymin, ymax);
    poly_aabb_at(sh, sB.c, sB.a, xmin, xmax, ymin, ymax);
    {
        float r2 = 0.0f;
        for (int i = 0; i < sh.n; ++i) { const V2 r = sh.v[i] - sh.centroid; r2 = fmaxf(r2, dot(r, r)); }
        const float da = sB.a - sB.a0, mrg = sqrtf(r2) * da * da * 0.125f + LINEAR_SLOP;
        xmin -= mrg; xmax += mrg; ymin -= mrg; ymax += mrg;
    }
    float body_min = 1.0f;
    for (int k = 0; k < cap; ++k) {
        Slot &sl = Cd.slot[base + k];
        if (sl.edge < 0 || (sl.toi_flags & 2) || sl.toi_count > MAX_SUB_STEPS) continue;
        if (!(sl.toi_flags & 1)) {
            Cd.slot_toi[base + k] = toi_alpha_terrain(M, Wd, Cd, bi, sl.edge, sB, xmin, xmax, ymin, ymax);
            sl.toi_flags |= 1;
        }
        body_min = fminf(body_min, Cd.slot_toi[base + k]);
    }
    S.body_minsep[bi] = body_min;
}

// b2World::SolveTOI.  `par`: the TOI of every candidate is computed by the lane that owns the body; the event loop itself
// (rare: a body arriving at the terrain within this step) runs on lane 0 of the env.
template <class Par>
MW_HD void solve_toi(const Model &M, Hot &Wd, Cold &Cd, Scratch &S, Par par, float h) {
    const int L0 = par.lane(), LN = par.n();
    const int NB = M.NB, NDP = M.n_dyn_pairs;
    // ---- pass 0 (by body): candidate contacts over the swept box (SynchronizeFixtures + FindNewContacts of the previous
    // step, done here where the sweep is known), flags reset, first time-of-impact of every contact
    for (int bi = L0; bi < NB; bi += LN) {
        const Shape &sh = M.shape[shape_of_body(bi)];
        S.body_minsep[bi] = 1.0f;
        if (sh.inv_mass == 0.0f) continue;  // static body
        const Sweep sB = sweep_of_body(M, Wd, Cd, bi);
        float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
        poly_aabb_at(sh, sB.c0, sB.a0, xmin, xmax, ymin, ymax);
        poly_aabb_at(sh, sB.c, sB.a, xmin, xmax, ymin, ymax);
        {   // a vertex leaves the box of its two end poses by at most |r| (1 - cos(da / 2)) <= |r| da^2 / 8 in between
            float r2 = 0.0f;
            for (int i = 0; i < sh.n; ++i) { const V2 r = sh.v[i] - sh.centroid; r2 = fmaxf(r2, dot(r, r)); }
            const float da = sB.a - sB.a0, mrg = sqrtf(r2) * da * da * 0.125f + LINEAR_SLOP;
            xmin -= mrg; xmax += mrg; ymin -= mrg; ymax += mrg;
        }
        int e0 = (int)floorf((xmin - 0.1f) / TERRAIN_STEP), e1 = (int)floorf((xmax + 0.1f) / TERRAIN_STEP);  // b2_aabbExtension
        if (e0 < 0) e0 = 0;
        if (e1 > M.NT - 2) e1 = M.NT - 2;
        Slot *slots = Cd.slot + M.slot_base[bi];
        const int cap = M.slot_cap[bi];
        for (int e = e0; e <= e1; ++e) {  // new candidates (not touching yet)
            Slot &sl = slots[e % cap];
            if (sl.edge < 0) { sl.edge = (int16_t)e; sl.npts = 0; sl.touching = 0; }
        }
        float body_min = 1.0f;
        for (int k = 0; k < cap; ++k) {
            Slot &sl = slots[k];
            sl.toi_flags = 0; sl.toi_count = 0;
            if (sl.edge < 0) continue;
            const float alpha = toi_alpha_terrain(M, Wd, Cd, bi, sl.edge, sB, xmin, xmax, ymin, ymax);
            Cd.slot_toi[M.slot_base[bi] + k] = alpha;
            sl.toi_flags = 1;
            body_min = fminf(body_min, alpha);
        }
        S.body_minsep[bi] = body_min;  // (the array is free after the position iterations) earliest event of this body
    }
    for (int p = L0; p < NDP; p += LN) {
        Slot &sl = Cd.slot[M.dyn_slot_base + p];
        sl.toi_flags = 0; sl.toi_count = 0;
        const int bA = M.dyn_a[p], bB = M.dyn_b[p];
        const bool stA = M.shape[shape_of_body(bA)].inv_mass == 0.0f, stB = M.shape[shape_of_body(bB)].inv_mass == 0.0f;
        if (stA == stB) continue;  // two dynamic (non-bullet) bodies: no continuous collision between them; two static: nothing moves
        Cd.slot_toi[M.dyn_slot_base + p] = toi_alpha_pair(M, Wd, Cd, bA, bB);
        sl.toi_flags = 1;
    }
    par.sync();
    if (L0 != 0 || M.continuous == 2) { par.sync(); return; }   // continuous == 2: timing experiments only (candidates without events)
    // ---- event loop (lane 0)
    for (int guard = 0; guard < 4 * MAX_TOI_CONTACTS; ++guard) {
        // the earliest event: per-body minima are kept in LDS (pass 0 / the end of the previous sub-step), only the winning
        // body's slots are looked at in HBM
        int min_slot = -1, min_bA = 0, min_bB = 0;
        float min_alpha = 1.0f;
        int min_body = -1;
        for (int bi = 0; bi < NB; ++bi) { const float a = S.body_minsep[bi]; if (a < min_alpha) { min_alpha = a; min_body = bi; } }
        if (min_body >= 0) {
            const int base = M.slot_base[min_body], cap = M.slot_cap[min_body];
            float best = 1.0f;
            for (int k = 0; k < cap; ++k) {
                const Slot &sl = Cd.slot[base + k];
                if (sl.edge < 0 || (sl.toi_flags & 2) || sl.toi_count > MAX_SUB_STEPS || !(sl.toi_flags & 1)) continue;
                const float alpha = Cd.slot_toi[base + k];
                if (alpha < best) { best = alpha; min_slot = base + k; min_bA = -1; min_bB = min_body; }
            }
            min_alpha = best;
        }
        for (int p = 0; p < NDP; ++p) {
            Slot &sl = Cd.slot[M.dyn_slot_base + p];
            const int bA = M.dyn_a[p], bB = M.dyn_b[p];
            const bool stA = M.shape[shape_of_body(bA)].inv_mass == 0.0f, stB = M.shape[shape_of_body(bB)].inv_mass == 0.0f;
            if (stA == stB || (sl.toi_flags & 2) || sl.toi_count > MAX_SUB_STEPS) continue;
            if (!(sl.toi_flags & 1)) { Cd.slot_toi[M.dyn_slot_base + p] = toi_alpha_pair(M, Wd, Cd, bA, bB); sl.toi_flags |= 1; }
            const float alpha = Cd.slot_toi[M.dyn_slot_base + p];
            if (alpha < min_alpha) { min_alpha = alpha; min_slot = M.dyn_slot_base + p; min_bA = bA; min_bB = bB; }
        }
        if (min_slot < 0 || 1.0f - 10.0f * B2_EPSILON < min_alpha) break;
        // ---- advance the moving body of the event to the time of impact (b2Body::Advance)
        const int mover = (min_bA >= 0 && M.shape[shape_of_body(min_bB)].inv_mass == 0.0f) ? min_bA : min_bB;  // the dynamic one
        const V2 bk_c0 = Cd.sweep_c0[mover], bk_c = Wd.b[mover].c;
        const float bk_a0 = Cd.sweep_a0[mover], bk_a = Wd.b[mover].a, bk_alpha0 = Cd.sweep_alpha0[mover];
        {
            Sweep sw = sweep_of_body(M, Wd, Cd, mover);
            sweep_advance(sw, min_alpha);
            Cd.sweep_c0[mover] = sw.c0; Cd.sweep_a0[mover] = sw.a0; Cd.sweep_alpha0[mover] = sw.alpha0;
            Wd.b[mover].c = sw.c0; Wd.b[mover].a = sw.a0;
        }
        Slot &ms = Cd.slot[min_slot];
        ManifoldOut mo;
        const bool touching = toi_update_contact(M, Wd, Cd, ms, min_bA, min_bB, mo);
        ms.toi_flags &= (uint8_t)~1u;
        ms.toi_count = (uint8_t)(ms.toi_count + 1);
        MW_STAT(toi_events, 1);
        if (!touching) {  // not solid after all: undo, and leave this contact alone for the rest of the step
            MW_STAT(toi_undone, 1);
            ms.toi_flags |= 2;
            Cd.sweep_c0[mover] = bk_c0; Cd.sweep_a0[mover] = bk_a0; Cd.sweep_alpha0[mover] = bk_alpha0;
            Wd.b[mover].c = bk_c; Wd.b[mover].a = bk_a;
            toi_refresh_body(M, Wd, Cd, S, mover);
            continue;
        }
        // ---- mini island: the event's contact and the mover's other touching contacts with static bodies
        int n_isl = 0;
        auto add_manifold = [&](const ManifoldOut &o, int bA, int bB, int slot_index, float friction) {
            if (n_isl >= MAX_TOI_CONTACTS || n_isl >= M.max_manifolds) return;
            Manifold &m = S.m[n_isl++];
            m.bA = (int8_t)bA; m.bB = (int8_t)bB; m.slot = (int16_t)slot_index; m.npts = (uint8_t)o.npts; m.type = (uint8_t)o.type;
            m.local_normal = o.local_normal; m.local_point = o.local_point;
            for (int i = 0; i < 2; ++i) { m.lp[i] = i < o.npts ? o.lp[i] : v2(0, 0); m.ni[i] = 0.0f; m.ti[i] = 0.0f; }  // no warm starting
            m.friction = friction;
        };
        const Shape &msh = M.shape[shape_of_body(mover)];
        if (min_bA < 0) add_manifold(mo, -1, mover, min_slot, sqrtf(FRICTION * msh.friction));
        else add_manifold(mo, min_bA, min_bB, min_slot, sqrtf(M.shape[shape_of_body(min_bA)].friction * M.shape[shape_of_body(min_bB)].friction));
        {
            const int base = M.slot_base[mover], cap = M.slot_cap[mover];
            float mv_x0 = 3.0e38f, mv_x1 = -3.0e38f, mv_y0 = 3.0e38f, mv_y1 = -3.0e38f;
            poly_aabb_at(msh, Wd.b[mover].c, Wd.b[mover].a, mv_x0, mv_x1, mv_y0, mv_y1);
            for (int k = 0; k < cap; ++k) {
                Slot &sl = Cd.slot[base + k];
                if (base + k == min_slot || sl.edge < 0) continue;
                if (!sl.touching) {  // an edge the body's box (at the time of impact) does not reach cannot start touching: nothing to update
                    const float ex0 = sl.edge * TERRAIN_STEP, ex1 = (sl.edge + 1) * TERRAIN_STEP;
                    const float ey0 = fminf(Cd.ty[sl.edge], Cd.ty[sl.edge + 1]), ey1 = fmaxf(Cd.ty[sl.edge], Cd.ty[sl.edge + 1]);
                    const float mg = 4.0f * POLY_RADIUS;
                    if (mv_x0 - mg > ex1 || mv_x1 + mg < ex0 || mv_y0 - mg > ey1 || mv_y1 + mg < ey0) continue;
                }
                ManifoldOut o2;
                if (toi_update_contact(M, Wd, Cd, sl, -1, mover, o2)) add_manifold(o2, -1, mover, base + k, sqrtf(FRICTION * msh.friction));
            }
        }
        // ---- b2Island::SolveTOI
        const MassAB qm = mass_of_pair(S, -1, mover);  // the static side carries no mass whichever slot it sits in
        auto mass_for = [&](const Manifold &m) -> MassAB {
            MassAB q = mass_of_pair(S, m.bA, m.bB);
            if (m.bA >= 0 && m.bA != mover) { q.mA = 0.0f; q.iA = 0.0f; }
            return q;
        };
        (void)qm;
        for (int it = 0; it < 20; ++it) {
            float ms_min = 0.0f;
            for (int k = 0; k < n_isl; ++k) {
                const Manifold &m = S.m[k];
                if (m.bB == mover) ms_min = fminf(ms_min, contact_solve_toi_position(Wd, m, mass_for(m)));
            }
            if (ms_min >= -1.5f * LINEAR_SLOP) break;
        }
        Cd.sweep_c0[mover] = Wd.b[mover].c; Cd.sweep_a0[mover] = Wd.b[mover].a;  // "leap of faith to new safe state"
        for (int k = 0; k < n_isl; ++k) contact_init_warm(Wd, S.m[k], mass_for(S.m[k]));  // impulses are zero: no warm start
        {
            // Box2D runs all the step's velocity iterations; once a whole sweep leaves every accumulated impulse and the body's
            // velocity exactly unchanged, every further sweep is the same no-op, so stopping there changes no bit of the result.
            // The moving body's velocity stays in registers over the sweeps (the static side never changes).
            V2 vB = Wd.b[mover].v, vA = v2(0, 0);
            float wB = Wd.b[mover].w, wA = 0.0f;
            // The sweep is a deterministic map of (velocity, accumulated impulses).  Besides the exact fixed point it often ends in a
            // short cycle (impulses flipping in their last bits): once state(i) == state(i - p), p <= 4, the state after the last of
            // the VEL_ITERS sweeps is known without running them.  History kept for islands of at most two manifolds.
            constexpr int NST = 3 + 4 * 2;
            float h1[NST], h2[NST], h3[NST], h4[NST];  // states after sweeps i - 1 .. i - 4
            for (int q = 0; q < NST; ++q) { h1[q] = 0.0f; h2[q] = 0.0f; h3[q] = 0.0f; h4[q] = 0.0f; }
            const bool track = n_isl <= 2;
            auto snapshot = [&](float *st) {
                st[0] = vB.x; st[1] = vB.y; st[2] = wB;
                for (int k = 0; k < 2; ++k) {
                    const bool on = k < n_isl;
                    st[3 + 4 * k] = on ? S.m[k].ni[0] : 0.0f; st[4 + 4 * k] = on ? S.m[k].ni[1] : 0.0f;
                    st[5 + 4 * k] = on ? S.m[k].ti[0] : 0.0f; st[6 + 4 * k] = on ? S.m[k].ti[1] : 0.0f;
                }
            };
            auto restore = [&](const float *st) {
                vB.x = st[0]; vB.y = st[1]; wB = st[2];
                for (int k = 0; k < 2; ++k) if (k < n_isl) { S.m[k].ni[0] = st[3 + 4 * k]; S.m[k].ni[1] = st[4 + 4 * k]; S.m[k].ti[0] = st[5 + 4 * k]; S.m[k].ti[1] = st[6 + 4 * k]; }
            };
            for (int it = 0; it < VEL_ITERS; ++it) {
                MW_STAT(toi_vel_iters, 1);
                for (int k = 0; k < n_isl; ++k) {
                    Manifold &m = S.m[k];
                    if (m.bB == mover) contact_solve_velocity_on(m, mass_for(m), vA, wA, vB, wB);
                    else { V2 z = v2(0, 0); float zw = 0.0f; contact_solve_velocity_on(m, mass_for(m), vB, wB, z, zw); }  // the mover is body A (known-answer scene only)
                }
                if (!track) { if (it == VEL_ITERS - 1) MW_STAT(toi_hist[9], 1); continue; }
                float cur[NST];
                snapshot(cur);
                bool same1 = it >= 1, same2 = it >= 2, same3 = it >= 3, same4 = it >= 4;
                for (int q = 0; q < NST; ++q) {
                    same1 = same1 && cur[q] == h1[q]; same2 = same2 && cur[q] == h2[q]; same3 = same3 && cur[q] == h3[q]; same4 = same4 && cur[q] == h4[q];
                }
                if (same1) { MW_STAT(toi_hist[it / 20], 1); break; }   // fixed point: every further sweep is a no-op
                const int period = same2 ? 2 : (same3 ? 3 : (same4 ? 4 : 0));
                if (period != 0) {  // state(j + period) = state(j) from here on; `left` sweeps remain: state(last) = state(it - period + left % period)
                    MW_STAT(toi_hist[8], 1);
                    const int left = (VEL_ITERS - 1 - it) % period;   // 0: cur
                    const int back = left == 0 ? 0 : period - left;    // the wanted state lies `back` sweeps before cur
                    if (back == 1) restore(h1); else if (back == 2) restore(h2); else if (back == 3) restore(h3);
                    break;
                }
                if (it == VEL_ITERS - 1) MW_STAT(toi_hist[9], 1);
                for (int q = 0; q < NST; ++q) { h4[q] = h3[q]; h3[q] = h2[q]; h2[q] = h1[q]; h1[q] = cur[q]; }
            }
            MW_STAT(toi_nisl[n_isl < 5 ? n_isl : 5], 1);
            Wd.b[mover].v = vB; Wd.b[mover].w = wB;
        }
        {   // integrate the rest of the step
            const float hs = (1.0f - min_alpha) * h;
            Body &b = Wd.b[mover];
            const V2 tr = hs * b.v;
            if (dot(tr, tr) > MAX_TRANSLATION * MAX_TRANSLATION) { const float ratio = MAX_TRANSLATION / sqrtf(dot(tr, tr)); b.v = ratio * b.v; }
            const float ro = hs * b.w;
            if (ro * ro > MAX_ROTATION * MAX_ROTATION) { const float ratio = MAX_ROTATION / fabsf(ro); b.w *= ratio; }
            b.c = b.c + hs * b.v;
            b.a += hs * b.w;
        }
        // the displaced body's cached times of impact are stale; its candidate set follows its new sweep (FindNewContacts)
        {
            const int base = M.slot_base[mover], cap = M.slot_cap[mover];
            for (int k = 0; k < cap; ++k) Cd.slot[base + k].toi_flags &= (uint8_t)~1u;
            float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
            poly_aabb_at(msh, Cd.sweep_c0[mover], Cd.sweep_a0[mover], xmin, xmax, ymin, ymax);
            poly_aabb_at(msh, Wd.b[mover].c, Wd.b[mover].a, xmin, xmax, ymin, ymax);
            int e0 = (int)floorf((xmin - 0.1f) / TERRAIN_STEP), e1 = (int)floorf((xmax + 0.1f) / TERRAIN_STEP);
            if (e0 < 0) e0 = 0;
            if (e1 > M.NT - 2) e1 = M.NT - 2;
            for (int e = e0; e <= e1; ++e) {
                Slot &sl = Cd.slot[base + e % cap];
                if (sl.edge < 0) { sl.edge = (int16_t)e; sl.npts = 0; sl.touching = 0; sl.toi_flags = 0; sl.toi_count = 0; }
            }
            for (int p = 0; p < NDP; ++p) if (M.dyn_a[p] == mover || M.dyn_b[p] == mover) Cd.slot[M.dyn_slot_base + p].toi_flags &= (uint8_t)~1u;
            toi_refresh_body(M, Wd, Cd, S, mover);
        }
    }
    par.sync();
}

// b2World::Step(1/50, 180, 60) for the lanes of `par`.
//
// Schedule of one Gauss-Seidel sweep (Box2D: all joints in creation order, then all contacts):
//   joints   three slots.  A walker's joints in creation order are hip0, knee0, hip1, knee1; hip0 -> knee0 share the
//            upper leg, hip0 -> hip1 the hull, hip1 -> knee1 the other upper leg, while knee0 and hip1 share nothing.
//            One lane per JOINT: slot 0 = hip0, slot 1 = knee0 | hip1, slot 2 = knee1, i.e. joint 4 w + 2 s + r (leg s,
//            r = 0 hip / 1 knee) runs in slot s + r; the three slots are one loop over the same lane-private JointCache.
//   contacts sub-slot i = the i-th terrain manifold of every body (by lane) and, while neither the package nor a hull
//            touches the terrain (`merge_ok`), the i-th active package-hull / hull-hull pair, solved by the lane that
//            owns the pair's second body -- leg-terrain and package-hull constraints share no body.  Otherwise the
//            dynamic pairs follow in sub-slots of their own, exactly the serial order.
template <class Par>
MW_HD_INLINE void world_step(const Model &M, Hot &Wd, Cold &Cd, Scratch &S, Par par) {
    const float h = 1.0f / FPS;
    const int L0 = par.lane(), LN = par.n();
    // the model scalars are read once: the solver loops below must not go back to memory for them
    const int NB = M.NB, NW = M.W, NDP = M.n_dyn_pairs, MAXMAN = M.max_manifolds;
    const int NODES = NW + 1;  // island graph nodes: walkers, then the package
    // lane-private constants of the bodies this lane owns (bi = L0 + kb * LN): manifold-list base, mass data, island node
    int own_sb[Par::BODIES], own_node[Par::BODIES];
    MassAB own_q[Par::BODIES];   // as body B of a terrain contact (A = terrain: zeros)
    MW_UNROLL
    for (int kb = 0; kb < Par::BODIES; ++kb) {
        const int bi = L0 + kb * LN;
        own_sb[kb] = 0; own_node[kb] = 0;
        own_q[kb].mA = 0.0f; own_q[kb].iA = 0.0f; own_q[kb].lcA = v2(0, 0); own_q[kb].mB = 0.0f; own_q[kb].iB = 0.0f; own_q[kb].lcB = v2(0, 0);
        if (bi >= NB) continue;
        const Shape &sh = M.shape[shape_of_body(bi)];
        own_q[kb].mB = sh.inv_mass; own_q[kb].iB = sh.inv_I; own_q[kb].lcB = sh.centroid;
        own_node[kb] = node_of(bi, NW);
        own_sb[kb] = M.slot_base[bi];
    }
    for (int sh = L0; sh < N_SHAPES; sh += LN) { S.sh_im[sh] = M.shape[sh].inv_mass; S.sh_ii[sh] = M.shape[sh].inv_I; S.sh_lc[sh] = M.shape[sh].centroid; }
    if (L0 == 0) S.nm = 0;
    par.sync();
    // ---- Collide: terrain candidates by body, then the dynamic pairs
    for (int bi = L0; bi < NB; bi += LN) collide_body_terrain(M, Wd, Cd, S, par, bi);
    par.sync();
    for (int p = L0; p < NDP; p += LN) collide_dyn_pair(M, Wd, Cd, S, par, p);
    par.sync();
    if (L0 == 0) {  // islands: walkers (+ package) joined by touching hull-hull / hull-package contacts; contact schedule
        for (int i = 0; i < NODES; ++i) { S.comp[i] = (int8_t)i; S.isl_done[i] = 0; }
        int nd = 0;
        for (int p = 0; p < NDP; ++p) {
            if (S.dyn_midx[p] < 0) continue;
            S.dyn_owner[nd] = (int8_t)M.dyn_b[p]; S.dyn_man[nd] = S.dyn_midx[p]; S.dyn_a_shape[nd] = (int8_t)shape_of_body(M.dyn_a[p]);
            S.dyn_list[nd++] = (int8_t)p;
            const int ca = S.comp[node_of(M.dyn_a[p], NW)], cb = S.comp[node_of(M.dyn_b[p], NW)];
            if (ca != cb) for (int i = 0; i < NODES; ++i) if (S.comp[i] == cb) S.comp[i] = (int8_t)ca;
        }
        S.n_dyn = (int8_t)nd;
        int mc = 0;
        bool merge = S.bm_cnt[0] == 0;
        for (int bi = 0; bi < NB; ++bi) {
            if (S.bm_cnt[bi] > mc) mc = S.bm_cnt[bi];
            if (bi >= 1 && (bi - 1) % 5 == 0 && S.bm_cnt[bi] != 0) merge = false;
        }
        S.max_cnt = (int8_t)mc; S.merge_ok = merge ? 1 : 0;
    }
    // ---- integrate velocities (gravity + the pending initial push)
    MW_UNROLL
    for (int kb = 0; kb < Par::BODIES; ++kb) {
        const int bi = L0 + kb * LN;
        if (bi >= NB) continue;
        Body &b = Wd.b[bi];
        float fx = 0.0f;
        if (bi >= 1 && (bi - 1) % 5 == 0) { const int w = (bi - 1) / 5; fx = Wd.push_x[w]; }
        const float im = own_q[kb].mB;
        if (im == 0.0f) continue;  // static body (b2Island::Solve integrates dynamic bodies only); the env has none
        b.v.x += h * (im * fx);
        b.v.y += h * (GRAVITY_Y + im * 0.0f);
        // linear/angular damping are 0: v *= 1/(1 + h*0)
    }
    par.sync();
    for (int w = L0; w < NW; w += LN) Wd.push_x[w] = 0.0f;  // ClearForces
    const int n_dyn = S.n_dyn;
    const int nsubA = S.merge_ok ? (S.max_cnt > n_dyn ? S.max_cnt : n_dyn) : S.max_cnt;
    const int nsubB = S.merge_ok ? 0 : n_dyn;
    const bool merge_ok = S.merge_ok != 0;
    MW_STAT(steps, 1); MW_STAT(sub_a, nsubA); MW_STAT(sub_b, nsubB); MW_STAT(manifolds, S.nm); MW_STAT(merged, merge_ok ? 1 : 0);
    // One contact sweep.  F_TERRAIN / F_DYN are statements over the manifold `m_`, the body `bi` (F_DYN also the pair `p_`).
    // MC: on the GPU every lane keeps ONE manifold of its first body in registers for the whole step -- its first terrain
    // manifold, or (a hull, while merge_ok) its package / hull pair -- so the common sub-slot needs no LDS look-ups.
    Manifold MC;
    int mc_idx = -1;
#define MW_MANIFOLD_DO(K_, F_) { const int k_ = (K_); if (Par::MCACHE && k_ == mc_idx) { Manifold &m_ = MC; F_; } else { Manifold &m_ = S.m[k_]; F_; } }
    /* a dynamic pair: B = this lane's body, A = the package (pair package-hull) or another hull */                            \
#define MW_DYN_MASS MassAB q_ = own_q[kb]; { const int sa_ = S.dyn_a_shape[i]; q_.mA = S.sh_im[sa_]; q_.iA = S.sh_ii[sa_]; q_.lcA = S.sh_lc[sa_]; }
#define MW_CONTACT_SWEEP(F_TERRAIN, F_DYN)                                                                   \
    for (int i = 0; i < nsubA; ++i) {                                                                        \
        const int dyn_own_ = (merge_ok && i < n_dyn) ? S.dyn_owner[i] : -1;                                  \
        MW_UNROLL                                                                                            \
        for (int kb = 0; kb < Par::BODIES; ++kb) {                                                           \
            const int bi = L0 + kb * LN;                                                                     \
            if (bi >= NB) continue;                                                                          \
            if (i < S.bm_cnt[bi]) { const MassAB &q_ = own_q[kb]; MW_MANIFOLD_DO(S.bm_idx[own_sb[kb] + i], F_TERRAIN) } \
            if (dyn_own_ == bi) { const int p_ = S.dyn_list[i]; (void)p_; MW_DYN_MASS MW_MANIFOLD_DO(S.dyn_man[i], F_DYN) } \
        }                                                                                                    \
        par.sync();                                                                                          \
    }                                                                                                        \
    for (int i = 0; i < nsubB; ++i) {                                                                        \
        const int dyn_own_ = S.dyn_owner[i];                                                                 \
        MW_UNROLL                                                                                            \
        for (int kb = 0; kb < Par::BODIES; ++kb) {                                                           \
            const int bi = L0 + kb * LN;                                                                     \
            if (bi < NB && dyn_own_ == bi) { const int p_ = S.dyn_list[i]; (void)p_; MW_DYN_MASS MW_MANIFOLD_DO(S.dyn_man[i], F_DYN) } \
        }                                                                                                    \
        par.sync();                                                                                          \
    }
    // ---- contact constraints: init + warm start
    MW_CONTACT_SWEEP(contact_init_warm(Wd, m_, q_), contact_init_warm(Wd, m_, q_))
    if (Par::MCACHE && L0 < NB) {
        if (S.bm_cnt[L0] > 0) mc_idx = S.bm_idx[own_sb[0]];
        else if (merge_ok) { for (int i = 0; i < n_dyn; ++i) if (S.dyn_owner[i] == L0) { mc_idx = S.dyn_man[i]; break; } }
        if (mc_idx >= 0) MC = S.m[mc_idx];
    }
    // ---- joints: init + warm start, in the three-slot order
    JointCache JC[Par::JOINTS];
    int jslot[Par::JOINTS];
    MW_UNROLL
    for (int kq = 0; kq < Par::JOINTS; ++kq) { const int ji = L0 + kq * LN; jslot[kq] = ji < 4 * NW ? ((ji >> 1) & 1) + (ji & 1) : -1; }
    for (int t = 0; t < 3; ++t) {
        MW_UNROLL
        for (int kq = 0; kq < Par::JOINTS; ++kq)
            if (jslot[kq] == t) joint_init_warm(M, Wd, Cd, S, L0 + kq * LN, h, JC[kq]);
        par.sync();
    }
    // ---- velocity iterations (islands are disjoint, so iterating them together changes nothing)
    for (int it = 0; it < VEL_ITERS; ++it) {
        for (int t = 0; t < 3; ++t) {
            MW_UNROLL
            for (int kq = 0; kq < Par::JOINTS; ++kq)
                if (jslot[kq] == t) joint_solve_velocity(Wd, JC[kq]);
            par.sync();
        }
        MW_CONTACT_SWEEP(contact_solve_velocity(Wd, m_, q_), contact_solve_velocity(Wd, m_, q_))
    }
    if (Par::MCACHE && mc_idx >= 0) { S.m[mc_idx].ni[0] = MC.ni[0]; S.m[mc_idx].ni[1] = MC.ni[1]; S.m[mc_idx].ti[0] = MC.ti[0]; S.m[mc_idx].ti[1] = MC.ti[1]; }
    // the accumulated joint impulses and limit states go back to the world (warm start of the next step)
    MW_UNROLL
    for (int kq = 0; kq < Par::JOINTS; ++kq) {
        if (jslot[kq] < 0) continue;
        Joint &j = Cd.j[L0 + kq * LN];
        j.ix = JC[kq].ix; j.iy = JC[kq].iy; j.iz = JC[kq].iz; j.motor_impulse = JC[kq].motor_impulse; j.limit_state = JC[kq].limit_state;
    }
    // ---- integrate positions
    for (int bi = L0; bi < NB; bi += LN) {
        Body &b = Wd.b[bi];
        V2 tr = h * b.v;
        if (dot(tr, tr) > MAX_TRANSLATION * MAX_TRANSLATION) { const float ratio = MAX_TRANSLATION / sqrtf(dot(tr, tr)); b.v = ratio * b.v; }
        const float ro = h * b.w;
        if (ro * ro > MAX_ROTATION * MAX_ROTATION) { const float ratio = MAX_ROTATION / fabsf(ro); b.w *= ratio; }
        Cd.sweep_c0[bi] = b.c; Cd.sweep_a0[bi] = b.a; Cd.sweep_alpha0[bi] = 0.0f;  // b2Island::Solve: sweep.c0 / a0 = the pose at the start of the step
        b.c = b.c + h * b.v;
        b.a += h * b.w;
    }
    par.sync();
    // ---- position iterations, each island stops on its own (b2Island::Solve early exit)
    for (int it = 0; it < POS_ITERS; ++it) {
        MW_STAT(pos_iters, 1);
        for (int bi = L0; bi < NB; bi += LN) S.body_minsep[bi] = 0.0f;
        for (int p = L0; p < NDYN; p += LN) S.dyn_minsep[p] = 0.0f;
        par.sync();
        MW_CONTACT_SWEEP(
            if (!S.isl_done[S.comp[own_node[kb]]]) S.body_minsep[bi] = fminf(S.body_minsep[bi], contact_solve_position(Wd, m_, q_)),
            if (!S.isl_done[S.comp[own_node[kb]]]) S.dyn_minsep[p_] = contact_solve_position(Wd, m_, q_))
        for (int w = L0; w < NW; w += LN) S.walker_ok[w] = 1;
        par.sync();
        for (int t = 0; t < 3; ++t) {
            MW_UNROLL
            for (int kq = 0; kq < Par::JOINTS; ++kq) {
                if (jslot[kq] != t) continue;
                const int w = (L0 + kq * LN) >> 2;
                if (S.isl_done[S.comp[w]]) continue;
                if (!joint_solve_position(Wd, JC[kq])) S.walker_ok[w] = 0;
            }
            par.sync();
        }
        if (L0 == 0) {
            bool all_done = true;
            for (int c = 0; c < NODES; ++c) {
                bool any = false;
                for (int i = 0; i < NODES; ++i) any |= (S.comp[i] == c);
                if (!any || S.isl_done[c]) continue;
                float ms = 0.0f;
                bool jok = true;
                for (int bi = 0; bi < NB; ++bi) if (S.comp[node_of(bi, NW)] == c) ms = fminf(ms, S.body_minsep[bi]);
                for (int i = 0; i < n_dyn; ++i) if (S.comp[node_of(S.dyn_owner[i], NW)] == c) ms = fminf(ms, S.dyn_minsep[S.dyn_list[i]]);
                for (int w = 0; w < NW; ++w) if (S.comp[w] == c) jok = jok && S.walker_ok[w];
                if (ms >= -3.0f * LINEAR_SLOP && jok) S.isl_done[c] = 1;
                else all_done = false;
            }
            S.all_done = all_done ? 1 : 0;
        }
        par.sync();
        if (S.all_done) break;
    }
    par.sync();
    // b2ContactSolver::StoreImpulses -> manifold cache (warm start of the next step)
    const int nm = S.nm < MAXMAN ? S.nm : MAXMAN;
    for (int k = L0; k < nm; k += LN) {
        const Manifold &m = S.m[k];
        Slot &sl = Cd.slot[m.slot];
        for (int i = 0; i < m.npts; ++i) { sl.ni[i] = m.ni[i]; sl.ti[i] = m.ti[i]; }
    }
    par.sync();
    // ---- continuous pass (b2World::Step: "if (m_continuousPhysics && step.dt > 0) SolveTOI(step)")
    if (M.continuous) solve_toi(M, Wd, Cd, S, par, h);
}
#undef MW_CONTACT_SWEEP
#undef MW_MANIFOLD_DO
#undef MW_DYN_MASS

// ---------------------------------------------------------------- lidar: b2EdgeShape::RayCast over the terrain
MW_HD float lidar_fraction(const Model &M, const Cold &Cd, V2 p1, V2 p2) {
    const V2 d = p2 - p1;
    float best = 1.0f;  // LidarCallback.fraction starts at 1.0 (:210)
    int e0 = (int)floorf(fminf(p1.x, p2.x) / TERRAIN_STEP), e1 = (int)floorf(fmaxf(p1.x, p2.x) / TERRAIN_STEP);
    if (e0 < 0) e0 = 0;
    if (e1 > M.NT - 2) e1 = M.NT - 2;
    for (int e = e0; e <= e1; ++e) {
        const V2 v1 = v2(e * TERRAIN_STEP, Cd.ty[e]), v2e = v2((e + 1) * TERRAIN_STEP, Cd.ty[e + 1]);
        const V2 ee = v2e - v1;
        V2 normal = v2(ee.y, -ee.x);
        { const float len = sqrtf(dot(normal, normal)); normal = (1.0f / len) * normal; }
        const float num = dot(normal, v1 - p1), den = dot(normal, d);
        if (den == 0.0f) continue;
        const float t = num / den;
        if (t < 0.0f || 1.0f < t) continue;
        const V2 q = p1 + t * d;
        const float rr = dot(ee, ee);
        if (rr == 0.0f) continue;
        const float s = dot(q - v1, ee) / rr;
        if (s < 0.0f || 1.0f < s) continue;
        if (t < best) best = t;
    }
    return best;
}

// ---------------------------------------------------------------- env: reset / step
struct EnvCfg {
    int n_walkers, reward_global, terminate_on_fall, one_hot, max_steps, auto_reset;
    float position_noise, angle_noise, forward_reward, fall_reward, drop_reward;
    uint32_t k0, k1;
};
constexpr int OBS_DIM = 24 + 4 + 3 + 1;  // :243
constexpr int MAX_AGENTS_ID = 40;      // MAX_AGENTS (:17): width of the one-hot id (:397-398)
MW_HD int obs_dim_of(const EnvCfg &C) { return OBS_DIM - 1 + (C.one_hot ? MAX_AGENTS_ID : 1); }  // :241-243

MW_HD void philox10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t o[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
enum : uint32_t { TAG_MW_TERRAIN = 32, TAG_MW_PUSH = 33, TAG_MW_NOISE = 34 };
MW_HD float u24f(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

MW_HD void env_observe(const Model &M, const EnvCfg &C, Hot &Wd, const Cold &Cd, uint32_t gid, float *obs, float *rew, uint8_t *done);

// MultiWalkerEnv.reset (:330-357) without its trailing step
MW_HD void env_reset_world(const Model &M, const EnvCfg &C, Hot &Wd, Cold &Cd, uint32_t gid) {
    const uint32_t tick = Wd.tick;
    Wd.game_over = 0; Wd.prev_package_shaping = 0.0f; Wd.t = 0;
    for (int w = 0; w < M.W; ++w) { Wd.fallen[w] = 0; Wd.prev_shaping[w] = 0.0f; Wd.ground[w][0] = Wd.ground[w][1] = 0; }
    for (int k = 0; k < MAXSLOT; ++k) { Cd.slot[k].edge = -1; Cd.slot[k].npts = 0; Cd.slot[k].touching = 0; }
    // _generate_terrain, non-hardcore branch (:516-612)
    {
        float velocity = 0.0f, y = TERRAIN_HEIGHT;
        int counter = TERRAIN_STARTPAD;
        bool oneshot = false;
        for (int i = 0; i < M.NT; ++i) {
            uint32_t r[4];
            philox10(gid, tick, (uint32_t)i, TAG_MW_TERRAIN, C.k0, C.k1, r);
            if (!oneshot) {
                const float sgn = (TERRAIN_HEIGHT - y) > 0.0f ? 1.0f : ((TERRAIN_HEIGHT - y) < 0.0f ? -1.0f : 0.0f);
                velocity = 0.8f * velocity + 0.01f * sgn;
                if (i > TERRAIN_STARTPAD) velocity += (2.0f * u24f(r[0]) - 1.0f) / SCALE;  // np_random.uniform(-1, 1) / SCALE
                y += velocity;
            }
            oneshot = false;
            Cd.ty[i] = y;
            counter -= 1;
            if (counter == 0) {
                counter = TERRAIN_GRASS / 2 + (int)(((uint64_t)r[1] * (uint64_t)(TERRAIN_GRASS - TERRAIN_GRASS / 2)) >> 32);  // randint(5, 10)
                oneshot = true;
            }
        }
    }
    // _generate_package (:499-514)
    float sx = 0.0f;
    for (int w = 0; w < M.W; ++w) sx += M.start_x[w];
    sx /= (float)M.W;
    {
        Body &b = Wd.b[0];
        b.a = 0.0f; b.v = v2(0, 0); b.w = 0.0f;
        b.c = v2(sx, TERRAIN_HEIGHT + 3 * LEG_H) + M.shape[SH_PACKAGE].centroid;
    }
    // BipedalWalker._reset (:113-192)
    const float init_y = TERRAIN_HEIGHT + 2 * LEG_H;
    for (int w = 0; w < M.W; ++w) {
        const float init_x = M.start_x[w];
        Body &hull = Wd.b[hull_of(w)];
        hull.a = 0.0f; hull.v = v2(0, 0); hull.w = 0.0f;
        hull.c = v2(init_x, init_y) + M.shape[SH_HULL].centroid;
        uint32_t r[4];
        philox10(gid, tick, (uint32_t)w, TAG_MW_PUSH, C.k0, C.k1, r);
        Wd.push_x[w] = (2.0f * u24f(r[0]) - 1.0f) * INITIAL_RANDOM;  // uniform(-INITIAL_RANDOM, INITIAL_RANDOM)
        for (int side = 0; side < 2; ++side) {
            const float sgn = side == 0 ? -1.0f : 1.0f;
            Body &up = Wd.b[hull_of(w) + 1 + 2 * side], &lo = Wd.b[hull_of(w) + 2 + 2 * side];
            up.a = sgn * 0.05f; up.v = v2(0, 0); up.w = 0.0f;
            up.c = v2(init_x, init_y - LEG_H / 2 - LEG_DOWN) + mul(rot(up.a), M.shape[SH_UPPER].centroid);
            lo.a = sgn * 0.05f; lo.v = v2(0, 0); lo.w = 0.0f;
            lo.c = v2(init_x, init_y - LEG_H * 3 / 2 - LEG_DOWN) + mul(rot(lo.a), M.shape[SH_LOWER].centroid);
            Joint &hip = Cd.j[4 * w + 2 * side], &knee = Cd.j[4 * w + 2 * side + 1];
            hip.ix = hip.iy = hip.iz = hip.motor_impulse = 0.0f; hip.limit_state = 0;
            hip.motor_speed = sgn; hip.max_torque = MOTORS_TORQUE;
            knee.ix = knee.iy = knee.iz = knee.motor_impulse = 0.0f; knee.limit_state = 0;
            knee.motor_speed = 1.0f; knee.max_torque = MOTORS_TORQUE;
        }
    }
    Wd.tick = tick + 1;
}

// MultiWalkerEnv.step (:359-428).  obs: [W][32], rew: [W]
template <class Par>
MW_HD_INLINE void env_step(const Model &M, const EnvCfg &C, Hot &Wd, Cold &Cd, Scratch &S, Par par, uint32_t gid, const float *actions, float *obs,
                    float *rew, uint8_t *done) {
    for (int w = par.lane(); w < M.W; w += par.n()) {  // apply_action (:194-203)
        for (int k = 0; k < 4; ++k) {
            const float a = actions[4 * w + k];
            Joint &j = Cd.j[4 * w + k];
            const float sp = (k % 2 == 0) ? SPEED_HIP : SPEED_KNEE;
            j.motor_speed = sp * (a > 0.0f ? 1.0f : (a < 0.0f ? -1.0f : 0.0f));
            j.max_torque = MOTORS_TORQUE * clampf(fabsf(a), 0.0f, 1.0f);
        }
    }
    par.sync();
    world_step(M, Wd, Cd, S, par);  // :365
    if (par.lane() == 0) {
        env_observe(M, C, Wd, Cd, gid, obs, rew, done);
        Wd.t += 1;
        Wd.tick += 1;
    }
    par.sync();
}

MW_HD void env_observe(const Model &M, const EnvCfg &C, Hot &Wd, const Cold &Cd, uint32_t gid, float *obs, float *rew, uint8_t *done) {
    const Body &pkg = Wd.b[0];
    const V2 pkg_pos = body_xf(M, pkg, 0).p;
    float rewards[MAX_WALKERS];
    V2 hull_pos[MAX_WALKERS];
    for (int w = 0; w < M.W; ++w) hull_pos[w] = body_xf(M, Wd.b[hull_of(w)], hull_of(w)).p;
    for (int w = 0; w < M.W; ++w) {
        const Body &hull = Wd.b[hull_of(w)];
        const V2 pos = hull_pos[w];
        float *o = obs + w * obs_dim_of(C);
        // get_observation (:205-237)
        o[0] = hull.a;
        o[1] = 2.0f * hull.w / FPS;
        o[2] = 0.3f * hull.v.x * (VIEWPORT_W / SCALE) / FPS;
        o[3] = 0.3f * hull.v.y * (VIEWPORT_H / SCALE) / FPS;
        for (int side = 0; side < 2; ++side) {
            const Body &up = Wd.b[hull_of(w) + 1 + 2 * side], &lo = Wd.b[hull_of(w) + 2 + 2 * side];
            o[4 + 5 * side + 0] = up.a - hull.a;                       // joints[0/2].angle
            o[4 + 5 * side + 1] = (up.w - hull.w) / SPEED_HIP;          // .speed / SPEED_HIP
            o[4 + 5 * side + 2] = (lo.a - up.a) + 1.0f;                 // joints[1/3].angle + 1.0
            o[4 + 5 * side + 3] = (lo.w - up.w) / SPEED_KNEE;
            o[4 + 5 * side + 4] = Wd.ground[w][side] ? 1.0f : 0.0f;
        }
        for (int i = 0; i < 10; ++i) {  // lidar (:209-214)
            float ls, lc;
            sincos_det(1.5f * i / 10.0f, ls, lc);
            const V2 p2 = v2(pos.x + ls * LIDAR_RANGE, pos.y - lc * LIDAR_RANGE);
            o[14 + i] = lidar_fraction(M, Cd, pos, p2);
        }
        // neighbours and package (:380-400), gaussian noise via Box-Muller on keyed uniforms
        float nz[7] = {0, 0, 0, 0, 0, 0, 0};
        if (C.position_noise != 0.0f || C.angle_noise != 0.0f) {
            for (int q = 0; q < 4; ++q) {
                uint32_t r[4];
                philox10(gid, Wd.tick, (uint32_t)(w * 4 + q), TAG_MW_NOISE, C.k0, C.k1, r);
                const float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f), u2 = u24f(r[1]);
                const float rad = sqrtf(-2.0f * logf(u1));
                float bs, bc;
                sincos_det(2.0f * B2_PI * u2, bs, bc);
                nz[2 * q] = rad * bc;
                if (2 * q + 1 < 7) nz[2 * q + 1] = rad * bs;
            }
        }
        int n = 24, zi = 0;
        for (int dj = -1; dj <= 1; dj += 2) {
            const int j = w + dj;
            if (j < 0 || j == M.W) { o[n++] = 0.0f; o[n++] = 0.0f; }
            else {
                const float xm = (hull_pos[j].x - pos.x) / M.package_length, ym = (hull_pos[j].y - pos.y) / M.package_length;
                o[n++] = xm + C.position_noise * nz[zi++];
                o[n++] = ym + C.position_noise * nz[zi++];
            }
        }
        const float xd = (pkg_pos.x - pos.x) / M.package_length, yd = (pkg_pos.y - pos.y) / M.package_length;
        o[n++] = xd + C.position_noise * nz[4];
        o[n++] = yd + C.position_noise * nz[5];
        o[n++] = pkg.a + C.angle_noise * nz[6];
        if (C.one_hot) { for (int k = 0; k < MAX_AGENTS_ID; ++k) o[n++] = (k == w) ? 1.0f : 0.0f; }  // np.eye(MAX_AGENTS)[i] :397-398
        else o[n++] = (float)w / (float)M.W;  // :400
        // shaping (:403-407)
        const float shaping = 0.0f - 5.0f * fabsf(o[0]);
        rewards[w] = shaping - Wd.prev_shaping[w];
        Wd.prev_shaping[w] = shaping;
    }
    const float package_shaping = C.forward_reward * 130.0f * pkg_pos.x / SCALE;  // :409-411
    for (int w = 0; w < M.W; ++w) rewards[w] += (package_shaping - Wd.prev_package_shaping);
    Wd.prev_package_shaping = package_shaping;
    bool dn = false;
    const float last_x = hull_pos[M.W - 1].x;  // `pos` leaks out of the loop: the LAST walker (:417, :420)
    if (Wd.game_over || last_x < 0.0f) { for (int w = 0; w < M.W; ++w) rewards[w] += C.drop_reward; dn = true; }
    if (last_x > (M.NT - TERRAIN_GRASS) * TERRAIN_STEP) dn = true;
    int nfallen = 0;
    for (int w = 0; w < M.W; ++w) { rewards[w] += C.fall_reward * (Wd.fallen[w] ? 1.0f : 0.0f); nfallen += Wd.fallen[w]; }
    if (C.terminate_on_fall && nfallen > 0) dn = true;
    if (rew) {
        if (C.reward_global) { float s = 0.0f; for (int w = 0; w < M.W; ++w) s += rewards[w]; s /= (float)M.W; for (int w = 0; w < M.W; ++w) rew[w] = s; }
        else for (int w = 0; w < M.W; ++w) rew[w] = rewards[w];
    }
    if (done) *done = dn ? 1 : 0;
}

}  // namespace mw
