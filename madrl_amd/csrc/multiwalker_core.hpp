// multiwalker_core.hpp -- MultiWalkerEnv dynamics: a from-scratch float32 restatement of the
// subset of Box2D 2.3.0 that madrl_environments/walker/multi_walker.py drives
// (`self.world.Step(1.0 / FPS, 6 * 30, 2 * 30)`, multi_walker.py:365), plus the env logic
// around it.  Written once as host/device code: the HIP kernels (multiwalker.hip) run it with the bodies,
// the terrain and the step's schedule in LDS and the contact cache in HBM.
//
// PARITY UNPINNED: the arithmetic lives in third-party Box2D (pybox2d /
// box2d-py, Box2D 2.3.0 inside; no version pin anywhere in the reference tree, the module is not
// installable here and the reference has no golden vectors at that boundary).  What follows
// restates Box2D's published algorithms from their documented structure -- and, since round 3, in
// Box2D's own ORDER, which is semantics for a Gauss-Seidel solver:
//   b2World::Step -> Collide (every contact: b2Contact::Update, feature-id matching, Begin / EndContact) ->
//   Solve (islands by depth-first search from the body list, newest body first, over the bodies' contact
//   edges -- newest contact first -- then their joint edges; b2Island::Solve with the island's joint and
//   contact order, per-island position-iteration exit and SLEEPING) -> SynchronizeFixtures (fat AABBs,
//   b2_aabbExtension / b2_aabbMultiplier) -> FindNewContacts (pairs created in (proxyIdA, proxyIdB) order, new
//   contacts go to the FRONT of the lists) -> SolveTOI.
//   b2ContactSolver (sequential impulses, block solver for 2-point manifolds, Baumgarte position correction);
//   b2RevoluteJoint (point constraint + motor + limit); b2CollidePolygons as of 2.3.0 (hill-climbing
//   b2FindMaxSeparation, 0.98 / 0.001 reference-face hysteresis); b2CollideEdgeAndPolygon on plain edges (the
//   reference builds edgeShape(vertices=[p1, p2]): no ghost vertices); b2EdgeShape::RayCast.
// The independent check of all of this is the test infrastructure's multiwalker_ref.c (plain C, Box2D's own data
// structures, no code shared with this file): tests/test_multiwalker_*.py compare the two step by step.
// What IS pinned to the reference (round 4): the ENV LAYER.  Everything multi_walker.py itself computes -- the world its reset()
// constructs call by call, the terrain walk, the pushes, apply_action, get_observation with the lidar callback, ContactDetector,
// the neighbour / package observations with their noise, rewards, termination, the one-hot id -- is recorded from the UNMODIFIED
// module running over a package named `Box2D` whose b2World is multiwalker_ref.c's (the test infrastructure's Box2D shim), and this file, free-running
// on the same terrain / pushes / actions, reproduces those recordings: body states bit for bit, float32 observations and rewards to
// 1e-6, flags and done exactly (tests/test_multiwalker_envlayer.py).  The dynamics (b2World::Step) stay restated, not run.
// Lane-parallel execution keeps Box2D's results: two constraints without a common body commute exactly, so any
// schedule that keeps, for every body, the island's order of the constraints touching it gives the bits of the
// serial sweep.  The solver runs on four lanes per env (build_islands: a walker's joints on its lane, the contacts
// dealt out by a list schedule of rounds x positions); the continuous pass runs every body's chain of events on
// its own lane and restores Box2D's world-wide event order afterwards (solve_toi).  The CPU build executes the
// solver lanes one after the other, in either order.
// Stated differences from the real library (same list as the oracle's header): D1 linear scan over
// fat AABBs instead of b2DynamicTree, proxy ids in creation order as in a fresh b2World; D2 lidar = closest hit;
// D3 Philox instead of numpy's Mersenne Twister; D4 a contact between two sleeping bodies is updated like any
// other (same manifold, the bodies have not moved; Box2D skips it); sin / cos from a +,-,* polynomial so that
// host and device builds agree bit for bit.
//
// Reference call sites (file:line in /root/reference/madrl_environments/walker/multi_walker.py):
//   constants :17-47 ; BipedalWalker._reset :113-192 ; apply_action :194-203 ;
//   get_observation :205-237 ; ContactDetector :50-84 ; MultiWalkerEnv.setup/reset :276-357 ;
//   step :359-428 ; _generate_package :499-514 ; _generate_terrain :516-628.
#pragma once

#include <math.h>
#include <string.h>
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

// One build of this source has room for MW_CAPW walkers and spreads an env over MW_NLANES lanes (multiwalker.hip is compiled once per
// capacity class -- 4 walkers on 4 lanes, 8 on 8, 10 on 16: the reference's curriculum runs n_walkers = 2 .. 10,
// lessons/multiwalker/env.yaml -- and the C ABI picks the class by n_walkers).  Every class lives in its own namespace, so that the
// three builds can be linked into one library; `mw` is an alias of the class the including file asked for.
#ifndef MW_CAPW
#define MW_CAPW 4
#endif
#ifndef MW_NLANES
#define MW_NLANES 4
#endif
#define MW_CAT2_(a, b) a##b
#define MW_CAT_(a, b) MW_CAT2_(a, b)
#define MW_NS MW_CAT_(mw_c, MW_CAPW)
namespace MW_NS {}
namespace mw = MW_NS;

namespace MW_NS {

// optional counters of the CPU build (scripts/mw_stats.cpp): how many sub-slots / position iterations a step really runs
#ifdef MW_STATS
struct Stats { long steps, sub_a, sub_b, manifolds, merged, pos_iters, toi_full, toi_culled, toi_events, toi_undone, toi_vel_iters, toi_hist[10], toi_nisl[6], cnt_hist[24], rounds_hist[12], toi_multi, toi_ties, toi_pairs, toi_hullpkg, lane_cost[4], cur_lane, pos_iters_step, maxcnt_step;
               long flops, phase, flops_by[6];
               int ev_n; unsigned char ev_body[64]; short ev_sweeps[64], ev_full[64]; long toi_full_chain; };   // the events of the current step (reset by the harness): body, velocity sweeps, full root-finder runs of the search before it   // phase: 0 collide, 1 solve, 2 broad phase after the solve, 3 time-of-impact search, 4 sub-steps, 5 observation   // float32 / float64 additions, multiplications, divisions, square roots: hand-counted per primitive (MW_FLOPS below)
extern Stats g_stats;
#define MW_STAT(f, v) (g_stats.f += (v))
inline int g_stats_lane();
#else
#define MW_STAT(f, v) ((void)0)
#endif
#ifdef MW_STATS
#define MW_FLOPS(n) (g_stats.flops += (n), g_stats.flops_by[g_stats.phase] += (n))   // the arithmetic of the primitive this sits in, counted by hand from its source
#define MW_PHASE(p) (g_stats.phase = (p))
#else
#define MW_FLOPS(n) ((void)0)
#define MW_PHASE(p) ((void)0)
#endif

// time stamps / per-env counters of measurement builds of the HIP kernels (scripts/mw_timing.py); nothing in every other build
#ifndef MW_TSTAMP
#define MW_TSTAMP(p, k) ((void)0)
#define MW_TVAL(p, k, v) ((void)0)
#endif
#ifndef MW_TACC
#define MW_TACC_T0(v) ((void)0)
#define MW_TACC(k, v) ((void)0)
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

constexpr int MAX_WALKERS = MW_CAPW;
static_assert(MAX_WALKERS >= 1 && MAX_WALKERS <= 12, "body / joint flag sets are at most 64 bits wide");
constexpr int MAXB = 5 * MAX_WALKERS + 1;        // package + 5 bodies per walker
constexpr int MAXJ = 4 * MAX_WALKERS;
constexpr int MAXT = TERRAIN_LENGTH * MAX_WALKERS / 8;  // terrain points (:301)
// contacts a body's cache holds (Model::slot_cap): the edges under its fat AABB (leg <= 6, hull <= 8 at the speeds of this env) PLUS the
// ones it has just left, which Box2D destroys only in the next step's Collide
constexpr int EDGE_SLOTS_LEG = 12, EDGE_SLOTS_HULL = 16;
// the package's cache: its length grows with the number of walkers (:293-294), and so does the run of edges under it
constexpr int package_slot_cap(int n_walkers) { return (int)(((float)(240.0 / 30.0 * (n_walkers / 1.75)) + 1.5f) / TERRAIN_STEP) + 12; }
constexpr int EDGE_SLOTS_PKG_MAX = (package_slot_cap(MAX_WALKERS) + 3) / 4 * 4;
static_assert(EDGE_SLOTS_PKG_MAX <= 128, "occupancy words of a contact cache");
constexpr int MAXSLOT = 4 * MAX_WALKERS * EDGE_SLOTS_LEG + MAX_WALKERS * EDGE_SLOTS_HULL + EDGE_SLOTS_PKG_MAX + MAX_WALKERS * (MAX_WALKERS - 1) / 2 + MAX_WALKERS;
// the active-manifold pool of a step (Model::max_manifolds <= MAXM): see build_model
constexpr int manifold_pool(int n_walkers) { return n_walkers <= 1 ? 20 : (n_walkers == 2 ? 28 : (n_walkers == 3 ? 36 : (n_walkers == 4 ? 40 : 6 * n_walkers + 16))); }
constexpr int MAXM = manifold_pool(MAX_WALKERS);
static_assert(MAXM <= 128, "manifold flag set of build_islands");
constexpr double TERRAIN_HEIGHT64 = 400.0 / 30.0 / 4, LEG_H64 = 34.0 / 30.0, LEG_DOWN64 = -8.0 / 30.0;   // the reference's float64 constants (:26-36)

struct V2 { float x, y; };
MW_HD V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
MW_HD V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
MW_HD V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
MW_HD V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
MW_HD V2 operator-(V2 a) { return v2(-a.x, -a.y); }
MW_HD float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
MW_HD float cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
MW_HD V2 cross(V2 a, float s) { return v2(s * a.y, -s * a.x); }
MW_HD V2 cross(float s, V2 a) { return v2(-s * a.y, s * a.x); }
// b2Min / b2Max / b2Clamp: plain comparisons (a < b ? a : b), whose result for signed zeros is the same on every target, unlike fminf / fmaxf
MW_HD float mnf(float a, float b) { return a < b ? a : b; }
MW_HD float mxf(float a, float b) { return a > b ? a : b; }
MW_HD float clampf(float a, float lo, float hi) { return mxf(lo, mnf(a, hi)); }
// Flag sets (one bit per body / joint / manifold / cache entry), as wide as the capacity class needs: one dword for four walkers -- the
// layout and the code of the rounds before the classes existed -- one or two qwords beyond.  No arrays: a word picked by a run-time index
// would be an indexed local object, which the GPU build keeps in scratch memory.
struct Bits32 {
    uint32_t v;
    MW_HD bool test(int i) const { return (v >> i) & 1u; }
    MW_HD void set(int i) { v |= 1u << i; }
    MW_HD void clear(int i) { v &= ~(1u << i); }
    MW_HD bool any() const { return v != 0; }
    MW_HD void join(Bits32 o) { v |= o.v; }
    MW_HD int pop_lowest() { const int k = __builtin_ctz(v); v &= v - 1; return k; }
    static MW_HD Bits32 none() { Bits32 b; b.v = 0; return b; }
    static MW_HD Bits32 first(int n) { Bits32 b; b.v = n >= 32 ? ~0u : (1u << n) - 1u; return b; }
};
struct Bits64 {
    uint64_t v;
    MW_HD bool test(int i) const { return (v >> i) & 1ull; }
    MW_HD void set(int i) { v |= 1ull << i; }
    MW_HD void clear(int i) { v &= ~(1ull << i); }
    MW_HD bool any() const { return v != 0; }
    MW_HD void join(Bits64 o) { v |= o.v; }
    MW_HD int pop_lowest() { const int k = __builtin_ctzll(v); v &= v - 1; return k; }
    static MW_HD Bits64 none() { Bits64 b; b.v = 0; return b; }
    static MW_HD Bits64 first(int n) { Bits64 b; b.v = n >= 64 ? ~0ull : (1ull << n) - 1ull; return b; }
};
struct Bits128 {
    uint64_t lo, hi;
    MW_HD bool test(int i) const { return ((i < 64 ? lo : hi) >> (i & 63)) & 1ull; }
    MW_HD void set(int i) { const uint64_t m = 1ull << (i & 63); if (i < 64) lo |= m; else hi |= m; }
    MW_HD void clear(int i) { const uint64_t m = ~(1ull << (i & 63)); if (i < 64) lo &= m; else hi &= m; }
    MW_HD bool any() const { return (lo | hi) != 0; }
    MW_HD void join(Bits128 o) { lo |= o.lo; hi |= o.hi; }
    MW_HD int pop_lowest() { if (lo != 0) { const int k = __builtin_ctzll(lo); lo &= lo - 1; return k; } const int k = __builtin_ctzll(hi); hi &= hi - 1; return 64 + k; }
    static MW_HD Bits128 none() { Bits128 b; b.lo = 0; b.hi = 0; return b; }
};
template <int N, bool A = (N <= 32), bool B = (N <= 64)> struct BitsFor { typedef Bits128 type; };
template <int N> struct BitsFor<N, true, true> { typedef Bits32 type; };
template <int N> struct BitsFor<N, false, true> { typedef Bits64 type; };
typedef BitsFor<MAXB>::type BodyBits;               // one bit per dynamic body
typedef BitsFor<MAXJ>::type JointBits;
typedef BitsFor<MAXM>::type ManifoldBits;           // ... per entry of the step's manifold pool
typedef BitsFor<EDGE_SLOTS_PKG_MAX <= 64 ? 64 : 128>::type SlotBits;   // ... per entry of one body's contact cache
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

}  // namespace MW_NS
#include "multiwalker_toi.hpp"
namespace MW_NS {


// ---------------------------------------------------------------- static model (per n_walkers)
constexpr float AABB_EXTENSION = 0.1f, AABB_MULTIPLIER = 2.0f;   // b2_aabbExtension, b2_aabbMultiplier
constexpr float TIME_TO_SLEEP = 0.5f, LINEAR_SLEEP_TOLERANCE = 0.01f, ANGULAR_SLEEP_TOLERANCE = 2.0f / 180.0f * B2_PI;
enum { SH_PACKAGE = 0, SH_HULL = 1, SH_UPPER = 2, SH_LOWER = 3, N_SHAPES = 4 };
struct Shape {
    int n;
    V2 v[5], nrm[5];
    V2 centroid;      // = body localCenter (b2PolygonShape::ComputeMass -> b2Body::ResetMassData; one fixture per body)
    V2 centroid_geo;  // b2PolygonShape::m_centroid (ComputeCentroid in Set, zero in SetAsBox): what the narrow phase uses
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
    int poly_rev;       // which b2CollidePolygons the hull / package pairs go through: 0 = Box2D 2.3.0 (default), 1 = later 2.3.x revisions
    Shape shape[N_SHAPES];
    JointDef jd[MAXJ];
    float package_length, package_scale;
    float start_x[MAX_WALKERS];
    double start_x64[MAX_WALKERS], mean_start_x64;
    double lidar_dx[10], lidar_dy[10];   // sin / cos(1.5 i / 10) * LIDAR_RANGE in float64 (:211-213)
    float tx[MAXT];       // terrain vertex x = float32(i * TERRAIN_STEP) with the float64 product of the reference (:521, :617)
    int slot_base[MAXB], slot_cap[MAXB];  // body-vs-terrain contact ranges in Cold::slot (contact with edge e lives in slot e % cap)
    uint8_t slot_body[MAXSLOT];           // the body whose cache holds terrain slot s
    int n_slots;                          // slots in use: the terrain caches and the pairs
    uint8_t toi_body[MAXB + MW_NLANES];    // the order in which the lanes take the bodies' event chains in the continuous pass (see build_model; 255: nobody)
    int n_toi;                            // entries of toi_body
    int dyn_slot_base, n_dyn_pairs;
    int dyn_a[MAX_WALKERS * (MAX_WALKERS - 1) / 2 + MAX_WALKERS], dyn_b[MAX_WALKERS * (MAX_WALKERS - 1) / 2 + MAX_WALKERS];
};
MW_HD int shape_of_body(int b) { return b == 0 ? SH_PACKAGE : (((b - 1) % 5 == 0) ? SH_HULL : (((b - 1) % 5) % 2 == 1 ? SH_UPPER : SH_LOWER)); }
MW_HD int hull_of(int w) { return 1 + 5 * w; }
MW_HD bool is_hull(int b) { return b >= 1 && (b - 1) % 5 == 0; }
// b2BroadPhase proxy ids, in creation order as in a fresh b2World (D1): package, the NT - 1 terrain edges, then the walkers' bodies
MW_HD int proxy_of_body(int b, int NT) { return b == 0 ? 0 : NT - 1 + b; }
MW_HD int proxy_of_edge(int e) { return 1 + e; }
// the order of the world's contact list: contacts are created batch by batch (one FindNewContacts call each), inside a batch in
// (proxyIdA, proxyIdB) order, and every new contact goes to the FRONT -- so a LARGER key is EARLIER in the list
MW_HD uint64_t contact_key(uint32_t batch, int pA, int pB) { return ((uint64_t)batch << 32) | ((uint64_t)(uint32_t)pA << 16) | (uint64_t)(uint32_t)pB; }

// b2PolygonShape::Set (gift wrapping from the right-most, lowest vertex, CCW) + normals + ComputeCentroid
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
        V2 nr = cross(e, 1.0f);                              // b2Cross(edge, 1.0f), then b2Vec2::Normalize: multiply by 1 / length
        const float inv = 1.0f / sqrtf(nr.x * nr.x + nr.y * nr.y);
        nr.x *= inv; nr.y *= inv;
        s.nrm[i] = nr;
    }
    {   // ComputeCentroid(m_vertices, m), reference point = origin
        V2 c = v2(0, 0);
        float area = 0.0f;
        const float inv3 = 1.0f / 3.0f;
        for (int i = 0; i < m; ++i) {
            const V2 p1 = v2(0, 0), p2 = s.v[i], p3 = i + 1 < m ? s.v[i + 1] : s.v[0];
            const V2 e1 = p2 - p1, e2 = p3 - p1;
            const float D = cross(e1, e2), ta = 0.5f * D;
            area += ta;
            c = c + (ta * inv3) * ((p1 + p2) + p3);
        }
        s.centroid_geo = (1.0f / area) * c;
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
    // b2Body::ResetMassData: localCenter = (mass * center) * (1 / mass); I -= mass * dot(localCenter, localCenter)
    const float inv_mass = 1.0f / mass;
    const V2 lc = inv_mass * (mass * c);
    Io -= mass * dot(lc, lc);
    s.centroid = lc;
    s.inv_mass = inv_mass;
    s.inv_I = 1.0f / Io;
}

inline void build_model(Model &M, int n_walkers) {
    M.W = n_walkers; M.NB = 5 * n_walkers + 1; M.NJ = 4 * n_walkers;
    M.continuous = 1;
    M.poly_rev = 0;
    // active-manifold pool: observed maxima of simultaneously touching pairs over long random / collapsed rollouts are 18, 25, 34
    // for 2, 3, 4 walkers and 30 .. 44 for 5 .. 10 (6 per walker + 16 there); a pair past the pool is ignored for the step and raises the
    // sticky Hot::overflow bit
    M.max_manifolds = manifold_pool(n_walkers);
    M.NT = (int)(TERRAIN_LENGTH * n_walkers * 1 / 8.0);          // :301
    const double scale64 = n_walkers / 1.75;                      // :293
    M.package_scale = (float)scale64;
    M.package_length = (float)(240.0 / 30.0 * scale64);           // :294
    const double step64 = 14.0 / 30.0, init_x64 = step64 * TERRAIN_STARTPAD / 2;
    double mean = 0.0;
    for (int w = 0; w < n_walkers; ++w) { M.start_x64[w] = init_x64 + WALKER_SEPERATION * w * step64; M.start_x[w] = (float)M.start_x64[w]; mean += M.start_x64[w]; }  // :285-287
    M.mean_start_x64 = mean / n_walkers;
    for (int i = 0; i < MAXT; ++i) M.tx[i] = (float)(i * step64);
    for (int i = 0; i < 10; ++i) { M.lidar_dx[i] = sin(1.5 * i / 10.0) * (160 / 30.0); M.lidar_dy[i] = cos(1.5 * i / 10.0) * (160 / 30.0); }
    {   // package :499-514: vertices (x * package_scale / SCALE, y / SCALE) in float64, then float32
        V2 p[4];
        const double px[4] = {-120, 120, 120, -120}, py[4] = {5, 5, -5, -5};
        for (int k = 0; k < 4; ++k) p[k] = v2((float)(px[k] * scale64 / 30.0), (float)(py[k] / 30.0));
        poly_set(M.shape[SH_PACKAGE], p, 4);
        poly_mass(M.shape[SH_PACKAGE], 1.0f);
        M.shape[SH_PACKAGE].friction = 0.5f; M.shape[SH_PACKAGE].category = 0x004; M.shape[SH_PACKAGE].mask = 0xFFFF;
    }
    {   // hull :118-127
        const double hx[5] = {-30, 6, 34, 34, -30}, hy[5] = {9, 9, 1, -8, -8};
        V2 p[5];
        for (int k = 0; k < 5; ++k) p[k] = v2((float)(hx[k] / 30.0), (float)(hy[k] / 30.0));
        poly_set(M.shape[SH_HULL], p, 5);
        poly_mass(M.shape[SH_HULL], 5.0f);
        M.shape[SH_HULL].friction = 0.1f; M.shape[SH_HULL].category = 0x002; M.shape[SH_HULL].mask = 0xFFFF;
    }
    for (int k = 0; k < 2; ++k) {  // legs :136-163: SetAsBox order, default friction 0.2
        Shape &s = M.shape[k == 0 ? SH_UPPER : SH_LOWER];
        const float hx = (float)((k == 0 ? 1.0 : 0.8) * (8.0 / 30.0) / 2), hy = (float)((34.0 / 30.0) / 2);
        s.n = 4;
        s.v[0] = v2(-hx, -hy); s.v[1] = v2(hx, -hy); s.v[2] = v2(hx, hy); s.v[3] = v2(-hx, hy);
        s.nrm[0] = v2(0, -1); s.nrm[1] = v2(1, 0); s.nrm[2] = v2(0, 1); s.nrm[3] = v2(-1, 0);
        s.centroid_geo = v2(0, 0);
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
    for (int b = 0; b < M.NB; ++b) {
        M.slot_base[b] = base;
        // candidate edges = those whose fat AABB overlaps the body's: a run no longer than (fat width + edge margins) / TERRAIN_STEP + 1
        M.slot_cap[b] = (b == 0) ? package_slot_cap(n_walkers) : (is_hull(b) ? EDGE_SLOTS_HULL : EDGE_SLOTS_LEG);
        base += M.slot_cap[b];
    }
    {   // Continuous pass: the k-th body of this list goes to lane k % lanes, and the lanes work through their bodies in lockstep -- all the
        // "first" bodies, then all the "second" ones ...  Each such group costs as much as its longest chain of events, so the bodies that
        // have events (lower legs, then upper legs) share groups instead of being spread over all of them; the package comes first in the
        // last group: lane 0, which has the contact cache sized for it (4W is a multiple of four lanes; with more lanes per env the list
        // is padded with empty entries up to the next multiple).
        int k = 0;
        for (int w = 0; w < n_walkers; ++w) { M.toi_body[k++] = (uint8_t)(hull_of(w) + 2); M.toi_body[k++] = (uint8_t)(hull_of(w) + 4); }
        for (int w = 0; w < n_walkers; ++w) { M.toi_body[k++] = (uint8_t)(hull_of(w) + 1); M.toi_body[k++] = (uint8_t)(hull_of(w) + 3); }
        while (k % MW_NLANES != 0) M.toi_body[k++] = 255;
        M.toi_body[k++] = 0;
        for (int w = 0; w < n_walkers; ++w) M.toi_body[k++] = (uint8_t)hull_of(w);
        M.n_toi = k;
    }
    M.dyn_slot_base = base;
    int np = 0;
    for (int w = 0; w < n_walkers; ++w) { M.dyn_a[np] = 0; M.dyn_b[np] = hull_of(w); ++np; }          // package (A, proxy 0) - hull
    for (int i = 0; i < n_walkers; ++i) for (int j = i + 1; j < n_walkers; ++j) { M.dyn_a[np] = hull_of(i); M.dyn_b[np] = hull_of(j); ++np; }
    M.n_dyn_pairs = np;
    M.n_slots = M.dyn_slot_base + np;
    for (int b = 0; b < M.NB; ++b) for (int k = 0; k < M.slot_cap[b]; ++k) M.slot_body[M.slot_base[b] + k] = (uint8_t)b;
}

// ---------------------------------------------------------------- dynamic state
struct Body { V2 c; float a; V2 v; float w; };  // centre of mass, angle, velocities
struct Joint {
    float ix, iy, iz, motor_impulse;  // accumulated impulses (warm start)
    float motor_speed, max_torque;
    int limit_state;                  // 0 inactive, 1 at lower, 2 at upper, 3 equal
};
struct Slot {       // one b2Contact between a fixed pair of fixtures; 32 bytes
    int16_t edge;   // terrain edge index (dyn pairs: 0), or -1: no contact (the fat AABBs do not overlap)
    uint8_t npts, touching;
    uint32_t id[2];
    float ni[2], ti[2];
    uint16_t batch;              // the FindNewContacts call that created it (contact_key): its place in Box2D's lists
    uint16_t reserved_;          // (e_toiFlag, m_toi, m_toiCount and e_enabledFlag only live inside one SolveTOI: Scratch::toi_*)
};
struct Manifold {   // one active b2ContactVelocityConstraint + b2ContactPositionConstraint
    int8_t bA, bB;         // bA = -1: static terrain
    int16_t slot;
    uint8_t npts, type, block, island;  // npts: points of the VELOCITY constraint (b2ContactSolver drops the second point of an ill-conditioned pair there only);
                                        // type bit 0: 0 faceA, 1 faceB, bits 1-2: points of the manifold = of the POSITION constraint; block: the 2-point block solver applies
    V2 local_normal, local_point, lp[2];  // b2Manifold (lp in the other body's frame)
    V2 normal, rA[2], rB[2];
    float friction, nm[2], tm[2], ni[2], ti[2];
    float k11, k12, k22, im11, im12, im22;  // block solver K and K^-1
};
static_assert(sizeof(Slot) == 32, "HBM record layout");
static_assert(sizeof(Manifold) == 140, "LDS budget of the HIP kernel");
// The env state is split by how often a step touches it.
//   Hot:  bodies and flags -- read and written by every one of the 180 + 60 solver sweeps; the HIP kernel keeps it in
//         LDS for the duration of the step.
//   Cold: the joints' persistent state, the contacts (warm-start impulses, list order), the broad phase's fat AABBs, sleep
//         times and the terrain heights -- touched once per step; the HIP kernel reads and writes it in place in HBM (L2).
struct Hot {
    Body b[MAXB];
    float push_x[MAX_WALKERS];    // ApplyForceToCenter pending until the first Step (:130-131)
    double prev_shaping[MAX_WALKERS], prev_package_shaping;   // float64 like the reference's Python side (:403-411)
    uint8_t fallen[MAX_WALKERS], ground[MAX_WALKERS][2], game_over, overflow;   // overflow: sticky; bit 0 a contact did not fit its cache / the pool, 1 a stale cache entry evicted, 2 the continuous pass's logs, 3 more FindNewContacts calls in one episode than Slot::batch counts
    uint8_t pad_[2 + (4 - (3 * MAX_WALKERS) % 4) % 4];
    uint32_t tick;                // observations of the current episode so far (noise draws)
    uint32_t episode;             // resets of this env so far (a reset's draws)
    uint32_t pad2_;
    int32_t t;
    BodyBits awake;               // bit b: body b is awake (b2Body::e_awakeFlag)
    uint32_t batch;               // FindNewContacts calls so far in this world (contact_key)
};
struct Cold {
    Joint j[MAXJ];                // warm-start impulses, motor targets, limit states: read at the start of a step into the
                                  // owning lane's JointCache, written back at its end
    float fat[MAXB][4];           // the broad phase's fat AABB of every dynamic body's proxy: lower x, y, upper x, y
    float sleep_time[MAXB];       // b2Body::m_sleepTime
    V2 sweep_c0[MAXB];            // b2Sweep::c0, a0 (the pose at the start of the step, later the last safe pose of the continuous pass)
    float sweep_a0[MAXB];
    float sweep_alpha0[MAXB];     // b2Sweep::alpha0
    float ty[MAXT];               // terrain heights, float32 as the b2EdgeShapes hold them (x = Model::tx)
    Slot slot[MAXSLOT];           // LAST member: a world of W walkers uses the first Model::n_slots of them (the HIP kernels that stage
                                  // this struct in LDS copy only that much)
};
struct World { Hot h; Cold c; };  // the packed per-env record in HBM
// How the code below sees a Cold record: one pointer per member, so that a kernel can keep some members in LDS and leave the others in HBM
// (the CPU build and the tests point all of them into one Cold).  The pointers are what is const in a `const ColdView &`, not the record.
struct ColdView {
    Joint *j;
    float (*fat)[4];
    float *sleep_time;
    V2 *sweep_c0;
    float *sweep_a0, *sweep_alpha0;
    float *ty;
    Slot *slot;
};
MW_HD ColdView cold_view(Cold &c) {
    ColdView v;
    v.j = c.j; v.fat = c.fat; v.sleep_time = c.sleep_time; v.sweep_c0 = c.sweep_c0; v.sweep_a0 = c.sweep_a0; v.sweep_alpha0 = c.sweep_alpha0; v.ty = c.ty; v.slot = c.slot;
    return v;
}

constexpr int NDYN = MAX_WALKERS * (MAX_WALKERS - 1) / 2 + MAX_WALKERS;
constexpr int MAXISL = MAX_WALKERS + 1;
// The island solver runs on SOLVE_LANES lanes per env: lane w owns the four joints of walker w, and the step's contacts are dealt out to
// the lanes by build_islands.  (The CPU build executes the lanes one after the other.)
constexpr int SOLVE_LANES = MW_NLANES;
static_assert(SOLVE_LANES >= MAX_WALKERS, "one solver lane per walker");
struct Scratch {  // per-step workspace (LDS on the GPU)
    // ---- what the solver launch keeps in LDS (up to `m_bA`)
    int nm;
    int8_t n_isl, n_rounds, max_cnt, pad_;
    BodyBits moved;            // bit b: body b's proxy is in the broad phase's move buffer
    BodyBits in_island;        // bit b: body b was simulated by this step's Solve (b2Body::e_islandFlag after b2World::Solve)
    // mass data of the four shapes (package, hull, upper leg, lower leg), copied once per step
    float sh_im[N_SHAPES], sh_ii[N_SHAPES];
    V2 sh_lc[N_SHAPES];
    // solver schedule (build_islands): per lane its manifolds in island order; manifold k runs in round m_round[k], inside a round in the
    // order of its position in the lane's list; per walker its joints in island order
    uint8_t lane_cnt[SOLVE_LANES], lane_list[SOLVE_LANES][MAXM];
    uint8_t m_round[MAXM];
    uint8_t jn[MAX_WALKERS], jorder[MAX_WALKERS][4];
    int8_t j_island[MAXJ];     // island of joint j, -1: none (its bodies are asleep and were not reached)
    int8_t island_of[MAXB];    // island of body b in this step's Solve, -1: none (asleep and not reached)
    uint8_t isl_pos_solved[MAXISL];   // b2Island::Solve's positionSolved of island c (the sleep test asks for it)
    // ---- the other launches
    // what the island construction needs of manifold k without going to the pool: its bodies and its place in Box2D's lists
    int8_t m_bA[MAXM], m_bB[MAXM];
    uint64_t m_key[MAXM];
#if MW_CAPW > 4
    // build_islands' bookkeeping.  With up to four walkers these are local arrays of the one lane that builds the islands, small enough for
    // the compiler to hold in registers; beyond, they would be scratch memory (tests/test_kernel_metadata.py), so they live here
    int8_t bi_lane[MAXB], bi_pos[MAXB], bi_round[MAXB], bi_lane_round[SOLVE_LANES];
    uint8_t bi_stack[MAXB + 2];
#endif
    alignas(16) Manifold m[MAXM];  // LAST member.  The CPU build's pool; the HIP kernels leave the pool in the state buffer (Model::max_manifolds entries)
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

// (written without a running output index: the small arrays of the narrow phase stay in registers on the GPU)
MW_HD int clip_segment(ClipV out[2], const ClipV in[2], V2 normal, float offset, int vertexIndexA) {
    const float d0 = dot(normal, in[0].v) - offset, d1 = dot(normal, in[1].v) - offset;
    const bool k0 = d0 <= 0.0f, k1 = d1 <= 0.0f, kx = d0 * d1 < 0.0f;   // k0 && k1 && kx cannot all hold
    ClipV x; x.v = v2(0, 0); x.id = 0;
    if (kx) {
        const float interp = d0 / (d0 - d1);
        x.v = in[0].v + interp * (in[1].v - in[0].v);
        x.id = mk_id(vertexIndexA, (in[0].id >> 8) & 0xFF, 0, 1);
    }
    // the kept points in the order in[0], in[1], intersection (field-wise selects on values, see sel())
    const ClipV a = in[0], b = in[1];
    out[0].v = sel(k0, a.v, sel(k1, b.v, x.v)); out[0].id = k0 ? a.id : (k1 ? b.id : x.id);
    out[1].v = sel(k0 && k1, b.v, x.v); out[1].id = (k0 && k1) ? b.id : x.id;
    return (k0 ? 1 : 0) + (k1 ? 1 : 0) + (kx ? 1 : 0);
}
struct ManifoldOut { int npts, type; V2 local_normal, local_point, lp[2]; uint32_t id[2]; };


// b2CollidePolygon.cpp as of Box2D 2.3.0: the separation of one edge normal of poly1 from poly2 ...
MW_HD float edge_separation(const Shape &p1, Xf xf1, int edge1, const Shape &p2, Xf xf2) {
    const V2 normal1_world = mul(xf1.q, p1.nrm[edge1]);
    const V2 normal1 = mulT(xf2.q, normal1_world);
    int index = 0;
    float min_dot = 3.402823466e+38f;
    for (int i = 0; i < p2.n; ++i) { const float d = dot(p2.v[i], normal1); if (d < min_dot) { min_dot = d; index = i; } }
    const V2 v1 = mul(xf1, p1.v[edge1]), v2w = mul(xf2, p2.v[index]);
    return dot(v2w - v1, normal1_world);
}
// ... and the hill-climbing search for the edge of maximum separation, started at the edge whose normal looks at poly2's centroid
MW_HD float find_max_separation(int *edge_index, const Shape &p1, Xf xf1, const Shape &p2, Xf xf2) {
    const int count1 = p1.n;
    const V2 d = mul(xf2, p2.centroid_geo) - mul(xf1, p1.centroid_geo);
    const V2 d_local1 = mulT(xf1.q, d);
    int edge = 0;
    float max_dot = -3.402823466e+38f;
    for (int i = 0; i < count1; ++i) { const float dt = dot(p1.nrm[i], d_local1); if (dt > max_dot) { max_dot = dt; edge = i; } }
    float s = edge_separation(p1, xf1, edge, p2, xf2);
    const int prev_edge = edge - 1 >= 0 ? edge - 1 : count1 - 1;
    const float s_prev = edge_separation(p1, xf1, prev_edge, p2, xf2);
    const int next_edge = edge + 1 < count1 ? edge + 1 : 0;
    const float s_next = edge_separation(p1, xf1, next_edge, p2, xf2);
    int best_edge, increment;
    float best_sep;
    if (s_prev > s && s_prev > s_next) { increment = -1; best_edge = prev_edge; best_sep = s_prev; }
    else if (s_next > s) { increment = 1; best_edge = next_edge; best_sep = s_next; }
    else { *edge_index = edge; return s; }
    for (;;) {
        if (increment == -1) edge = best_edge - 1 >= 0 ? best_edge - 1 : count1 - 1;
        else edge = best_edge + 1 < count1 ? best_edge + 1 : 0;
        s = edge_separation(p1, xf1, edge, p2, xf2);
        if (s > best_sep) { best_edge = edge; best_sep = s; }
        else break;
    }
    *edge_index = best_edge;
    return best_sep;
}

// ... and b2FindMaxSeparation as LATER 2.3.x revisions have it: every edge normal of poly1, taken to poly2's frame, against the deepest
// vertex of poly2 -- no hill climbing, no centroids (Model::poly_rev = 1; which revision the authors' pybox2d wrapped is not recorded)
MW_HD float find_max_separation_all_edges(int *edge_index, const Shape &p1, Xf xf1, const Shape &p2, Xf xf2) {
    const Xf xf = mulT(xf2, xf1);
    int best = 0;
    float max_sep = -3.402823466e+38f;
    for (int i = 0; i < p1.n; ++i) {
        const V2 n = mul(xf.q, p1.nrm[i]), v1 = mul(xf, p1.v[i]);
        float si = 3.402823466e+38f;
        for (int j = 0; j < p2.n; ++j) { const float sij = dot(n, p2.v[j] - v1); if (sij < si) si = sij; }
        if (si > max_sep) { max_sep = si; best = i; }
    }
    *edge_index = best;
    return max_sep;
}

// b2CollidePolygons (rev 0 = 2.3.0: hill-climbing search, reference face chosen with the 0.98 / 0.001 hysteresis; rev 1 = later 2.3.x:
// exhaustive search, poly1 = B only beyond 0.1 * b2_linearSlop)
MW_HD void collide_polygons(ManifoldOut &mo, const Shape &pA, Xf xfA, const Shape &pB, Xf xfB, int rev) {
    MW_FLOPS(420);   // b2CollidePolygons: two b2FindMaxSeparation searches, incident edge, two clips
    mo.npts = 0;
    const float total_radius = 2.0f * POLY_RADIUS;
    int edgeA = 0, edgeB = 0;
    const float sepA = rev ? find_max_separation_all_edges(&edgeA, pA, xfA, pB, xfB) : find_max_separation(&edgeA, pA, xfA, pB, xfB);
    if (sepA > total_radius) return;
    const float sepB = rev ? find_max_separation_all_edges(&edgeB, pB, xfB, pA, xfA) : find_max_separation(&edgeB, pB, xfB, pA, xfA);
    if (sepB > total_radius) return;
    const Shape *p1, *p2; Xf xf1, xf2; int edge1, flip;
    const float k_relative_tol = 0.98f, k_absolute_tol = 0.001f;
    if (rev ? (sepB > sepA + 0.1f * LINEAR_SLOP) : (sepB > k_relative_tol * sepA + k_absolute_tol)) { p1 = &pB; p2 = &pA; xf1 = xfB; xf2 = xfA; edge1 = edgeB; mo.type = 1; flip = 1; }
    else { p1 = &pA; p2 = &pB; xf1 = xfA; xf2 = xfB; edge1 = edgeA; mo.type = 0; flip = 0; }
    ClipV inc[2];
    {   // b2FindIncidentEdge
        const V2 normal1 = mulT(xf2.q, mul(xf1.q, p1->nrm[edge1]));
        int index = 0; float mind = 3.402823466e+38f;
        for (int i = 0; i < p2->n; ++i) { const float dd = dot(normal1, p2->nrm[i]); if (dd < mind) { mind = dd; index = i; } }
        const int i1 = index, i2 = (i1 + 1 < p2->n) ? i1 + 1 : 0;
        inc[0].v = mul(xf2, p2->v[i1]); inc[0].id = mk_id(edge1, i1, 1, 0);
        inc[1].v = mul(xf2, p2->v[i2]); inc[1].id = mk_id(edge1, i2, 1, 0);
    }
    const int iv1 = edge1, iv2 = (edge1 + 1 < p1->n) ? edge1 + 1 : 0;
    V2 v11 = p1->v[iv1], v12 = p1->v[iv2];
    V2 local_tangent = v12 - v11;
    { const float inv = 1.0f / sqrtf(dot(local_tangent, local_tangent)); local_tangent.x *= inv; local_tangent.y *= inv; }   // b2Vec2::Normalize
    const V2 local_normal = cross(local_tangent, 1.0f), plane_point = 0.5f * (v11 + v12);
    const V2 tangent = mul(xf1.q, local_tangent), normal = cross(tangent, 1.0f);
    v11 = mul(xf1, v11); v12 = mul(xf1, v12);
    const float front_offset = dot(normal, v11);
    const float side1 = -dot(tangent, v11) + total_radius, side2 = dot(tangent, v12) + total_radius;
    ClipV c1[2], c2[2];
    if (clip_segment(c1, inc, -tangent, side1, iv1) < 2) return;
    if (clip_segment(c2, c1, tangent, side2, iv2) < 2) return;
    mo.local_normal = local_normal; mo.local_point = plane_point;
    V2 cand[2]; uint32_t cid[2]; bool keep[2];   // the kept clip points, packed to the front without a running index
    MW_UNROLL
    for (int i = 0; i < 2; ++i) {
        const float sep = dot(normal, c2[i].v) - front_offset;
        keep[i] = sep <= total_radius;
        cand[i] = mulT(xf2, c2[i].v);
        cid[i] = flip ? swap_id(c2[i].id) : c2[i].id;
    }
    // fixed destinations, selected VALUES (entries past npts are never read)
    mo.lp[0] = sel(keep[0], cand[0], cand[1]); mo.id[0] = keep[0] ? cid[0] : cid[1];
    mo.lp[1] = cand[1]; mo.id[1] = cid[1];
    mo.npts = (keep[0] ? 1 : 0) + (keep[1] ? 1 : 0);
}


// b2CollideEdgeAndPolygon (b2EPCollider::Collide); edge = shape A, whose body frame is the world.  has0 / has3: ghost vertices
// (b2EdgeShape::m_hasVertex0 / 3); the terrain edges of the reference have none.
MW_HD void collide_edge_polygon(ManifoldOut &mo, V2 v1, V2 v2e, const Shape &pB, Xf xfB, bool has0, V2 v0, bool has3, V2 v3) {
    mo.npts = 0;
    const Xf xf = xfB;  // edge body transform is identity
    const V2 centroidB = mul(xf, pB.centroid_geo);
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
    V2x5 bv, bn;   // (named members, see V2x5)
    const int nB = pB.n;
#define MW_E5(F_) F_(0, e0) F_(1, e1) F_(2, e2) F_(3, e3) F_(4, e4)
#define MW_XF_(i, e) bv.e = v2(0, 0); bn.e = v2(0, 0); if (i < nB) { bv.e = mul(xf, pB.v[i]); bn.e = mul(xf.q, pB.nrm[i]); }
    MW_E5(MW_XF_)
#undef MW_XF_
    const float radius = 2.0f * POLY_RADIUS;
    // edge axis
    float edge_sep = 3.0e38f;
#define MW_ES_(i, e) if (i < nB) { const float s = dot(m_normal, bv.e - v1); if (s < edge_sep) edge_sep = s; }
    MW_E5(MW_ES_)
#undef MW_ES_
    if (edge_sep > radius) return;
    // polygon axis
    int poly_type = 0, poly_index = -1; float poly_sep = -3.0e38f;
    const V2 perp = v2(-m_normal.y, m_normal.x);
    bool stop = false;   // Box2D leaves the loop at the first separating axis
#define MW_PA_(i, e) if (i < nB && !stop) {                                                                     \
        const V2 n = -bn.e;                                                                                    \
        const float s1 = dot(n, bv.e - v1), s2 = dot(n, bv.e - v2e), s = mnf(s1, s2);                          \
        if (s > radius) { poly_type = 2; poly_index = i; poly_sep = s; stop = true; }                          \
        else {                                                                                                 \
            bool skip;                                                                                         \
            if (dot(n, perp) >= 0.0f) skip = dot(n - upper, m_normal) < -ANGULAR_SLOP;                         \
            else skip = dot(n - lower, m_normal) < -ANGULAR_SLOP;                                              \
            if (!skip && s > poly_sep) { poly_type = 2; poly_index = i; poly_sep = s; }                        \
        }                                                                                                      \
    }
    MW_E5(MW_PA_)
#undef MW_PA_
    if (poly_type != 0 && poly_sep > radius) return;
    const float k_rel = 0.98f, k_abs = 0.001f;
    const bool primary_edge = (poly_type == 0) || !(poly_sep > k_rel * edge_sep + k_abs);
    ClipV ie[2];
    int rf_i1, rf_i2; V2 rf_v1, rf_v2, rf_normal;
    if (primary_edge) {
        mo.type = 0;
        int best = 0; float bestv = dot(m_normal, bn.e0);
#define MW_BI_(i, e) if (i >= 1 && i < nB) { const float val = dot(m_normal, bn.e); if (val < bestv) { bestv = val; best = i; } }
        MW_E5(MW_BI_)
#undef MW_BI_
#undef MW_E5
        const int i1 = best, i2 = (i1 + 1 < nB) ? i1 + 1 : 0;
        ie[0].v = pick(bv, i1); ie[0].id = mk_id(0, i1, 1, 0);
        ie[1].v = pick(bv, i2); ie[1].id = mk_id(0, i2, 1, 0);
        if (front) { rf_i1 = 0; rf_i2 = 1; rf_v1 = v1; rf_v2 = v2e; rf_normal = normal1; }
        else { rf_i1 = 1; rf_i2 = 0; rf_v1 = v2e; rf_v2 = v1; rf_normal = -normal1; }
    } else {
        mo.type = 1;
        ie[0].v = v1; ie[0].id = mk_id(0, poly_index, 0, 1);
        ie[1].v = v2e; ie[1].id = mk_id(0, poly_index, 0, 1);
        rf_i1 = poly_index; rf_i2 = (rf_i1 + 1 < nB) ? rf_i1 + 1 : 0;
        rf_v1 = pick(bv, rf_i1); rf_v2 = pick(bv, rf_i2); rf_normal = pick(bn, rf_i1);
    }
    const V2 side_n1 = v2(rf_normal.y, -rf_normal.x), side_n2 = -side_n1;
    const float so1 = dot(side_n1, rf_v1), so2 = dot(side_n2, rf_v2);
    ClipV c1[2], c2[2];
    if (clip_segment(c1, ie, side_n1, so1, rf_i1) < 2) return;
    if (clip_segment(c2, c1, side_n2, so2, rf_i2) < 2) return;
    if (primary_edge) { mo.local_normal = rf_normal; mo.local_point = rf_v1; }
    else { mo.local_normal = pB.nrm[rf_i1]; mo.local_point = pB.v[rf_i1]; }
    // the kept clip points, packed to the front (no running index: ManifoldOut stays in registers)
    V2 cand[2]; uint32_t cid[2]; bool keep[2];
    MW_UNROLL
    for (int i = 0; i < 2; ++i) {
        const float sep = dot(rf_normal, c2[i].v - rf_v1);
        keep[i] = sep <= radius;
        if (primary_edge) { cand[i] = mulT(xf, c2[i].v); cid[i] = c2[i].id; }
        else { cand[i] = c2[i].v; cid[i] = swap_id(c2[i].id); }
    }
    // fixed destinations, selected VALUES (entries past npts are never read)
    mo.lp[0] = sel(keep[0], cand[0], cand[1]); mo.id[0] = keep[0] ? cid[0] : cid[1];
    mo.lp[1] = cand[1]; mo.id[1] = cid[1];
    mo.npts = (keep[0] ? 1 : 0) + (keep[1] ? 1 : 0);
}


// ---------------------------------------------------------------- broad phase: fat AABBs (b2DynamicTree node boxes; D1)
struct AABB { float lx, ly, hx, hy; };
MW_HD AABB poly_aabb(const Shape &s, Xf t) {   // b2PolygonShape::ComputeAABB
    V2 lower = mul(t, s.v[0]), upper = lower;
    for (int i = 1; i < s.n; ++i) {
        const V2 p = mul(t, s.v[i]);
        lower = v2(mnf(lower.x, p.x), mnf(lower.y, p.y));
        upper = v2(mxf(upper.x, p.x), mxf(upper.y, p.y));
    }
    AABB b; b.lx = lower.x - POLY_RADIUS; b.ly = lower.y - POLY_RADIUS; b.hx = upper.x + POLY_RADIUS; b.hy = upper.y + POLY_RADIUS;
    return b;
}
MW_HD AABB fatten(AABB a) { AABB f; f.lx = a.lx - AABB_EXTENSION; f.ly = a.ly - AABB_EXTENSION; f.hx = a.hx + AABB_EXTENSION; f.hy = a.hy + AABB_EXTENSION; return f; }
// terrain edge e: b2EdgeShape::ComputeAABB at the identity transform, fattened at proxy creation; static, so it never changes
MW_HD AABB edge_fat_aabb(const Model &M, const ColdView &Cd, int e) {
    const float x1 = M.tx[e], x2 = M.tx[e + 1], y1 = Cd.ty[e], y2 = Cd.ty[e + 1];
    AABB a; a.lx = mnf(x1, x2) - POLY_RADIUS; a.ly = mnf(y1, y2) - POLY_RADIUS; a.hx = mxf(x1, x2) + POLY_RADIUS; a.hy = mxf(y1, y2) + POLY_RADIUS;
    return fatten(a);
}
MW_HD bool aabb_overlap(const AABB &a, const AABB &b) {   // b2TestOverlap
    if (b.lx - a.hx > 0.0f || b.ly - a.hy > 0.0f) return false;
    if (a.lx - b.hx > 0.0f || a.ly - b.hy > 0.0f) return false;
    return true;
}
MW_HD bool aabb_contains(const AABB &a, const AABB &b) { return a.lx <= b.lx && a.ly <= b.ly && b.hx <= a.hx && b.hy <= a.hy; }
MW_HD AABB body_fat(const ColdView &Cd, int b) { AABB f; f.lx = Cd.fat[b][0]; f.ly = Cd.fat[b][1]; f.hx = Cd.fat[b][2]; f.hy = Cd.fat[b][3]; return f; }
MW_HD void set_body_fat(const ColdView &Cd, int b, const AABB &f) { Cd.fat[b][0] = f.lx; Cd.fat[b][1] = f.ly; Cd.fat[b][2] = f.hx; Cd.fat[b][3] = f.hy; }
// edges whose fat AABB can overlap [lx, hx] in x: a conservative index range, every candidate is then tested exactly
MW_HD void edge_range(const Model &M, float lx, float hx, int &e0, int &e1) {
    e0 = (int)floorf((lx - 0.12f) / TERRAIN_STEP) - 1;
    e1 = (int)floorf((hx + 0.12f) / TERRAIN_STEP) + 1;
    if (e0 < 0) e0 = 0;
    if (e1 > M.NT - 2) e1 = M.NT - 2;
}

// ContactDetector.BeginContact (:56-78) for a pair that started touching, except the lower legs' ground_contact, which also
// EndContact writes and which therefore depends on the ORDER of the events (see collide_body_terrain).  bA == -1: terrain.
MW_HD void contact_begin_flags(Hot &Wd, int bA, int bB) {
    if (bA >= 0 && is_hull(bA) && bB != 0) Wd.fallen[(bA - 1) / 5] = 1;   // a hull touches anything but the package
    if (is_hull(bB) && bA != 0) Wd.fallen[(bB - 1) / 5] = 1;
    if (bA == 0 && !is_hull(bB)) Wd.game_over = 1;                        // the package touches a non-hull
    if (bB == 0 && !(bA >= 0 && is_hull(bA))) Wd.game_over = 1;
}
MW_HD bool is_lower_leg(int b) { return b >= 1 && ((b - 1) % 5 == 2 || (b - 1) % 5 == 4); }
MW_HD void set_ground_flag(Hot &Wd, int b, bool on) { Wd.ground[(b - 1) / 5][(b - 1) % 5 == 2 ? 0 : 1] = on ? 1 : 0; }

// ---------------------------------------------------------------- lane parallelism
// The step is written once for a group of cooperating lanes (`Par`): SerialPar (CPU build: one lane that owns every body,
// no-op sync, and executes the four solver lanes one after the other) or a group of four lanes of a wavefront in the HIP
// kernels (16 envs per wavefront).  Bodies by lane (body b on lane b % n) wherever the work is per body; the solver's and
// the continuous pass's own mappings are described where they are built (build_islands, Model::toi_body).
struct SerialPar {
    static constexpr int SOLVE_EMU = SOLVE_LANES;   // solver lanes this thread executes, one after the other
    static constexpr int MREG = 2;                  // manifolds a solver lane keeps in lane-private storage (the rest: the pool)
    static constexpr int SOLVE_OVERFLOW = 3;        // room for that many manifolds in the overflow copies (the HIP launch: LDS)
    Manifold ovf[SOLVE_OVERFLOW];
    MW_HD Manifold *solve_overflow() { return ovf; }
    bool rev = false;                               // test hook: run the solver lanes in descending order
    MW_HD int lane() const { return 0; }
    MW_HD int n() const { return 1; }
    MW_HD int solve_lane(int i) const { return rev ? SOLVE_LANES - 1 - i : i; }
    MW_HD void sync() const {}
    MW_HD int alloc(int *counter) const { return (*counter)++; }
    MW_HD void or_bits(uint32_t *p, uint32_t v) const { *p |= v; }
    template <class B> MW_HD void or_bits(B *p, B v) const { p->join(v); }
    MW_HD uint32_t reduce_or(uint32_t v) const { return v; }   // OR over the lanes of the env: this one lane has seen everything
    // (the HIP build of the sixteen-lane class finds its record again from the lane id here, see GroupPar in multiwalker_impl.hpp)
    MW_HD ColdView cold_again(const ColdView &c) const { return c; }
    MW_HD Manifold *pool_again(Manifold *p) const { return p; }
};

// b2Contact::Update of one contact whose new manifold is `mo`: impulses carried over by feature id, touching state, e_enabledFlag
// set again.  Returns 0: no event, 1: BeginContact, 2: EndContact.
MW_HD int contact_update(Slot &sl, const ManifoldOut &mo) {
    // new point i takes the impulses of the FIRST old point with its feature id (two points at most on either side: written out, so that
    // nothing here is an indexed array on the GPU)
    const int on = sl.npts, nn = mo.npts;
    const uint32_t o0 = sl.id[0], o1 = sl.id[1];
    const float sn0 = sl.ni[0], sn1 = sl.ni[1], st0 = sl.ti[0], st1 = sl.ti[1];
    float ni0 = 0.0f, ni1 = 0.0f, ti0 = 0.0f, ti1 = 0.0f;
    if (nn > 0) { if (on > 0 && o0 == mo.id[0]) { ni0 = sn0; ti0 = st0; } else if (on > 1 && o1 == mo.id[0]) { ni0 = sn1; ti0 = st1; } }
    if (nn > 1) { if (on > 0 && o0 == mo.id[1]) { ni1 = sn0; ti1 = st0; } else if (on > 1 && o1 == mo.id[1]) { ni1 = sn1; ti1 = st1; } }
    const bool touching = nn > 0, was = sl.touching != 0;
    sl.touching = touching ? 1 : 0; sl.npts = (uint8_t)nn;
    if (nn > 0) { sl.id[0] = mo.id[0]; sl.ni[0] = ni0; sl.ti[0] = ti0; }
    if (nn > 1) { sl.id[1] = mo.id[1]; sl.ni[1] = ni1; sl.ti[1] = ti1; }
    return touching == was ? 0 : (touching ? 1 : 2);
}
// a touching contact becomes a solver manifold (pool slot from par.alloc; the solver's order is decided later by build_islands)
template <class Par>
MW_HD int emit_manifold(Hot &Wd, Scratch &S, Manifold *MP, Par par, const Slot &sl, int slot_index, uint64_t key, const ManifoldOut &mo, int bA, int bB, float friction, int max_manifolds) {
    const int idx = par.alloc(&S.nm);
    if (idx >= max_manifolds) { Wd.overflow |= 1; return -1; }  // pool exhausted: the pair is ignored this step (sticky flag)
    S.m_bA[idx] = (int8_t)bA; S.m_bB[idx] = (int8_t)bB; S.m_key[idx] = key;
    Manifold &m = MP[idx];
    m.bA = (int8_t)bA; m.bB = (int8_t)bB; m.slot = (int16_t)slot_index; m.npts = (uint8_t)mo.npts; m.type = (uint8_t)(mo.type | (mo.npts << 1)); m.island = 0;
    m.local_normal = mo.local_normal; m.local_point = mo.local_point;
    // (component by component: a struct copy out of a conditionally written local is done with integer loads, and those keep it in memory)
    if (mo.npts > 0) { m.lp[0].x = mo.lp[0].x; m.lp[0].y = mo.lp[0].y; m.ni[0] = sl.ni[0]; m.ti[0] = sl.ti[0]; }
    if (mo.npts > 1) { m.lp[1].x = mo.lp[1].x; m.lp[1].y = mo.lp[1].y; m.ni[1] = sl.ni[1]; m.ti[1] = sl.ti[1]; }
    m.friction = friction;
    return idx;
}
MW_HD void edge_polygon_manifold(const Model &M, const ColdView &Cd, int e, const Shape &s, Xf xfB, ManifoldOut &mo) {
    MW_FLOPS(260);   // b2CollideEdgeAndPolygon: polygon into the edge's frame, edge and polygon separations, reference face, two clips
    const V2 p1 = v2(M.tx[e], Cd.ty[e]), p2 = v2(M.tx[e + 1], Cd.ty[e + 1]);
    collide_edge_polygon(mo, p1, p2, s, xfB, false, p1, false, p2);   // plain b2EdgeShape: no ghost vertices (:617-620)
}

// Which entries of a body's contact cache hold a contact (bit k: slots[k].edge >= 0).  The cache lives in HBM in the HIP kernels: the
// loads of one group of eight are independent, so a scan costs a few memory round trips instead of one per entry, and the callers then
// touch only the occupied entries (a handful out of 12 - 44).
MW_HD SlotBits occupied_slots(const Slot *slots, int cap) {
    SlotBits m = SlotBits::none();
    for (int k = 0; k < cap; k += 8) {
        int e[8];
        MW_UNROLL
        for (int q = 0; q < 8; ++q) e[q] = k + q < cap ? (int)slots[k + q].edge : -1;
        MW_UNROLL
        for (int q = 0; q < 8; ++q) if (e[q] >= 0) m.set(k + q);
    }
    return m;
}

// b2ContactManager::Collide for the contacts of body `bi` with terrain edges.  Box2D walks the world's contact list (newest
// first); the only thing that order decides here is a lower leg's ground_contact when one pass holds both a Begin and an End
// for it: the LAST event of the walk -- the one on the OLDEST contact -- wins.
template <class Par>
MW_HD void collide_body_terrain(const Model &M, Hot &Wd, const ColdView &Cd, Scratch &S, Manifold *MP, Par par, int bi) {
    const Shape &s = M.shape[shape_of_body(bi)];
    const AABB fatb = body_fat(Cd, bi);
    Slot *slots = Cd.slot + M.slot_base[bi];
    const int cap = M.slot_cap[bi];
    const Xf xfB = body_xf(M, Wd.b[bi], bi);
    const AABB tight = poly_aabb(s, xfB);
    const int pb = proxy_of_body(bi, M.NT);
    uint64_t ev_key = ~0ull;
    int ev_kind = 0;
    const float fr = sqrtf(FRICTION * s.friction);   // b2MixFriction
    for (SlotBits occ = occupied_slots(slots, cap); occ.any();) {
        const int k = occ.pop_lowest();
        Slot &sl = slots[k];
        const int e = sl.edge;
        // (0, edge) for the package, (edge, body) for a walker's body: b2Contact::Create stores the edge as fixture A either way
        const uint64_t key = bi == 0 ? contact_key(sl.batch, 0, proxy_of_edge(e)) : contact_key(sl.batch, proxy_of_edge(e), pb);
        int ev;
        if (!aabb_overlap(edge_fat_aabb(M, Cd, e), fatb)) {   // the fat AABBs ceased to overlap: b2ContactManager::Destroy
            ev = sl.touching ? 2 : 0;
            sl.edge = -1; sl.npts = 0; sl.touching = 0;
        } else {
            ManifoldOut mo; mo.npts = 0;
            // cull (never changes a result): the tight boxes are further apart than any manifold reaches
            const float elo = mnf(Cd.ty[e], Cd.ty[e + 1]), ehi = mxf(Cd.ty[e], Cd.ty[e + 1]);
            if (!(tight.ly > ehi + 0.1f || tight.hy < elo - 0.1f || tight.lx > M.tx[e + 1] + 0.1f || tight.hx < M.tx[e] - 0.1f))
                edge_polygon_manifold(M, Cd, e, s, xfB, mo);
            ev = contact_update(sl, mo);
            if (ev == 1) contact_begin_flags(Wd, -1, bi);
            if (sl.touching) emit_manifold(Wd, S, MP, par, sl, M.slot_base[bi] + k, key, mo, -1, bi, fr, M.max_manifolds);
        }
        if (ev != 0 && key < ev_key) { ev_key = key; ev_kind = ev; }
    }
    if (ev_kind != 0 && is_lower_leg(bi)) set_ground_flag(Wd, bi, ev_kind == 1);
}

// package - hull and hull - hull pair p (two dynamic bodies: filtered by b2ContactFilter, never jointed)
template <class Par>
MW_HD void collide_dyn_pair(const Model &M, Hot &Wd, const ColdView &Cd, Scratch &S, Manifold *MP, Par par, int p) {
    const int bA = M.dyn_a[p], bB = M.dyn_b[p];
    Slot &sl = Cd.slot[M.dyn_slot_base + p];
    if (sl.edge < 0) return;
    if (!aabb_overlap(body_fat(Cd, bA), body_fat(Cd, bB))) {   // Destroy (an EndContact would only clear lower-leg flags: none here)
        // b2Contact::Destroy: "if (manifold.pointCount > 0 && no sensor) bodyA->SetAwake(true), bodyB->SetAwake(true)" -- a package thrown off
        // a sleeping walker's hull within one step wakes that hull (once in ~10^7 env-steps: found by scripts/mw_soak.py --seed 1000 at six walkers)
        if (sl.touching) {
            BodyBits wake = BodyBits::none();
            for (int q = 0; q < 2; ++q) { const int b = q ? bB : bA; if (!Wd.awake.test(b)) { wake.set(b); Cd.sleep_time[b] = 0.0f; } }
            if (wake.any()) par.or_bits(&Wd.awake, wake);
        }
        sl.edge = -1; sl.npts = 0; sl.touching = 0;
        return;
    }
    const Shape &sA = M.shape[shape_of_body(bA)], &sB = M.shape[shape_of_body(bB)];
    ManifoldOut mo; mo.npts = 0;
    const Xf xfA = body_xf(M, Wd.b[bA], bA), xfB = body_xf(M, Wd.b[bB], bB);
    const AABB ta = poly_aabb(sA, xfA), tb = poly_aabb(sB, xfB);
    if (!(ta.lx > tb.hx + 0.1f || tb.lx > ta.hx + 0.1f || ta.ly > tb.hy + 0.1f || tb.ly > ta.hy + 0.1f)) collide_polygons(mo, sA, xfA, sB, xfB, M.poly_rev);
    const bool was = sl.touching != 0;
    const int ev = contact_update(sl, mo);
    if (ev == 1) contact_begin_flags(Wd, bA, bB);
    if (ev != 0) {   // "if (touching != wasTouching) bodyA->SetAwake(true), bodyB->SetAwake(true)"
        // (the pairs are dealt over the env's lanes: the flag word is OR-ed atomically; two lanes waking the same body do the same thing)
        BodyBits wake = BodyBits::none();
        for (int q = 0; q < 2; ++q) { const int b = q ? bB : bA; if (!Wd.awake.test(b)) { wake.set(b); Cd.sleep_time[b] = 0.0f; } }
        if (wake.any()) par.or_bits(&Wd.awake, wake);
    }
    (void)was;
    if (sl.touching)
        emit_manifold(Wd, S, MP, par, sl, M.dyn_slot_base + p, contact_key(sl.batch, proxy_of_body(bA, M.NT), proxy_of_body(bB, M.NT)), mo, bA, bB,
                      sqrtf(sA.friction * sB.friction), M.max_manifolds);
}

// b2Body::SynchronizeFixtures -> b2Fixture::Synchronize -> b2BroadPhase::MoveProxy: the proxy's box is the union of the boxes at
// the sweep's start pose (c0, a0) and at the current pose; it is re-fattened (extension + twice the displacement) and buffered as
// moved only when it left its fat AABB.
MW_HD bool sync_fixture(const Model &M, const Hot &Wd, const ColdView &Cd, int b) {
    MW_FLOPS(150);
    const Shape &s = M.shape[shape_of_body(b)];
    const Xf xf1 = xf_from(Cd.sweep_c0[b], Cd.sweep_a0[b], s.centroid), xf2 = body_xf(M, Wd.b[b], b);
    const AABB a1 = poly_aabb(s, xf1), a2 = poly_aabb(s, xf2);
    AABB a; a.lx = mnf(a1.lx, a2.lx); a.ly = mnf(a1.ly, a2.ly); a.hx = mxf(a1.hx, a2.hx); a.hy = mxf(a1.hy, a2.hy);
    if (aabb_contains(body_fat(Cd, b), a)) return false;
    AABB f = fatten(a);
    const V2 d = AABB_MULTIPLIER * (xf2.p - xf1.p);   // predict AABB displacement
    if (d.x < 0.0f) f.lx += d.x; else f.hx += d.x;
    if (d.y < 0.0f) f.ly += d.y; else f.hy += d.y;
    set_body_fat(Cd, b, f);
    return true;
}
// b2ContactManager::FindNewContacts for body b's moved proxy against the terrain: a contact is created (e_enabledFlag set, no
// points) for every edge whose fat AABB overlaps and that has none yet
MW_HD void find_new_terrain_contacts(const Model &M, Hot &Wd, const ColdView &Cd, int b, uint32_t batch, uint16_t tag = 0) {
    const AABB fatb = body_fat(Cd, b);
    int e0, e1;
    edge_range(M, fatb.lx, fatb.hx, e0, e1);
    Slot *slots = Cd.slot + M.slot_base[b];
    const int cap = M.slot_cap[b];
    for (int e = e0; e <= e1; ++e) {
        if (!aabb_overlap(edge_fat_aabb(M, Cd, e), fatb)) continue;
        Slot &sl = slots[e % cap];
        if (sl.edge == e) continue;                    // the contact exists
        if (sl.edge >= 0) {
            // The cache slot holds another edge.  If that contact is not touching and its edge has left the body's new fat AABB, Box2D
            // would destroy it in the next Collide and nothing can observe it before (it cannot be hit inside a fat AABB that holds the
            // whole sweep): it is dropped now (sticky bit 1).  Otherwise the new pair is ignored (sticky bit 0).
            if (!sl.touching && !aabb_overlap(edge_fat_aabb(M, Cd, sl.edge), fatb)) Wd.overflow |= 2;
            else { Wd.overflow |= 1; continue; }
        }
        sl.edge = (int16_t)e; sl.npts = 0; sl.touching = 0; sl.batch = (uint16_t)batch; sl.reserved_ = tag;
    }
}
// (one lane runs this; b2ContactManager::AddPair of Box2D 2.3.0 ends with "Wake up the bodies": a hull put to sleep by this very step's
// islands, or a sleeping package, is awake again when a moved proxy's fat AABB starts to overlap its own -- the terrain's pairs need nothing:
// their dynamic body is the one that moved.  Found by scripts/mw_soak.py --seed 1000 at six walkers, env-step 127 913 of that run.)
MW_HD void find_new_pair_contacts(const Model &M, Hot &Wd, const ColdView &Cd, BodyBits moved, uint32_t batch) {
    for (int p = 0; p < M.n_dyn_pairs; ++p) {
        const int bA = M.dyn_a[p], bB = M.dyn_b[p];
        Slot &sl = Cd.slot[M.dyn_slot_base + p];
        if (sl.edge >= 0 || !(moved.test(bA) || moved.test(bB))) continue;
        if (!aabb_overlap(body_fat(Cd, bA), body_fat(Cd, bB))) continue;
        sl.edge = 0; sl.npts = 0; sl.touching = 0; sl.batch = (uint16_t)batch;
        for (int q = 0; q < 2; ++q) { const int b = q ? bB : bA; if (!Wd.awake.test(b)) { Wd.awake.set(b); Cd.sleep_time[b] = 0.0f; } }
    }
}

// ---------------------------------------------------------------- islands (b2World::Solve) and the solver schedule
// The key of the contact in slot `si` seen from anywhere (its place in the world list and in both bodies' edge lists).
MW_HD uint64_t slot_key(const Model &M, const ColdView &Cd, int si) {
    const Slot &sl = Cd.slot[si];
    if (si >= M.dyn_slot_base) { const int p = si - M.dyn_slot_base; return contact_key(sl.batch, proxy_of_body(M.dyn_a[p], M.NT), proxy_of_body(M.dyn_b[p], M.NT)); }
    int b = 0;
    while (b + 1 < M.NB && si >= M.slot_base[b + 1]) ++b;
    return b == 0 ? contact_key(sl.batch, 0, proxy_of_edge(sl.edge)) : contact_key(sl.batch, proxy_of_edge(sl.edge), proxy_of_body(b, M.NT));
}
// The next entry of body b's contact-edge list after the one with key `below` (newest first = descending key), among the contacts
// that are touching and have a manifold (every contact is enabled when Solve runs: Collide has just updated it); returns the manifold
// or -1.  (b2World::Solve: "for (b2ContactEdge* ce = b->m_contactList; ce; ce = ce->next)".)
MW_HD int next_contact_edge(const Scratch &S, int nm, int b, uint64_t below, uint64_t &key_out) {
    int best = -1;
    uint64_t bk = 0;
    for (int k = 0; k < nm; ++k) {
        if (S.m_bA[k] != b && S.m_bB[k] != b) continue;
        const uint64_t key = S.m_key[k];
        if (key < below && (best < 0 || key > bk)) { best = k; bk = key; }
    }
    key_out = bk;
    return best;
}

// b2World::Solve's island construction, run by ONE lane: seeds in body-list order (last created body first), depth-first search
// over contact edges then joint edges.  In that order the constraints get their place in the solver's schedule: a walker's joints on its
// lane, every contact on the lane that can run it earliest (rounds x positions).  A sleeping seed is skipped; every body reached is woken.
MW_HD void build_islands(const Model &M, Hot &Wd, const ColdView &Cd, Scratch &S, Manifold *MP) {
    const int NB = M.NB, NW = M.W;
    const int nm = S.nm < M.max_manifolds ? S.nm : M.max_manifolds;
    // the last contact constraint scheduled on body b: its lane, position in that lane's list and round (lane -1: none yet)
#if MW_CAPW > 4
    int8_t (&b_lane)[MAXB] = S.bi_lane, (&b_pos)[MAXB] = S.bi_pos, (&b_round)[MAXB] = S.bi_round;
    int8_t (&lane_round)[SOLVE_LANES] = S.bi_lane_round;
    uint8_t (&stack)[MAXB + 2] = S.bi_stack;
#else
    int8_t b_lane[MAXB], b_pos[MAXB], b_round[MAXB];
    int8_t lane_round[SOLVE_LANES];
    int stack[MAXB + 2];
#endif
    BodyBits flag = BodyBits::none();           // e_islandFlag of the bodies
    JointBits jflag = JointBits::none();        // of the joints
    ManifoldBits cflag = ManifoldBits::none();  // of the manifolds (pool index)
    for (int b = 0; b < NB; ++b) { S.island_of[b] = -1; b_lane[b] = -1; b_pos[b] = 0; b_round[b] = 0; }
    for (int j = 0; j < 4 * NW; ++j) S.j_island[j] = -1;
    for (int l = 0; l < SOLVE_LANES; ++l) { S.lane_cnt[l] = 0; lane_round[l] = 0; }
    for (int w = 0; w < NW; ++w) S.jn[w] = 0;
    int n_isl = 0, max_round = -1, max_cnt = 0;
    for (int s = 0; s < NB; ++s) {
        const int seed = s < NB - 1 ? NB - 1 - s : 0;   // body list: the walkers' bodies newest first, (static terrain,) the package last
        if (flag.test(seed)) continue;
        if (!Wd.awake.test(seed)) continue;
        const int isl = n_isl++;
        S.isl_pos_solved[isl] = 0;
        int sp = 0;
        stack[sp++] = seed;
        flag.set(seed);
        while (sp > 0) {
            const int b = stack[--sp];
            S.island_of[b] = (int8_t)isl;
            if (!Wd.awake.test(b)) { Wd.awake.set(b); Cd.sleep_time[b] = 0.0f; }   // "make sure the body is awake"
            uint64_t below = ~0ull, key;
            for (int mi = next_contact_edge(S, nm, b, below, key); mi >= 0; mi = next_contact_edge(S, nm, b, below, key)) {
                below = key;
                if (cflag.test(mi)) continue;     // already in an island (reached from its other body)
                cflag.set(mi);
                struct { int bA, bB; } m = {S.m_bA[mi], S.m_bB[mi]};
                MP[mi].island = (uint8_t)isl;
                // schedule: a sweep runs round by round and inside a round position by position of the lanes' lists, all lanes at once.
                // Any lane may hold any contact (the bodies live in shared memory): the contact goes to the lane that can run it earliest,
                // where it must sit lexicographically after the last constraint of either body that another lane holds.
                int ln = 0, pos = 0, rd = 0, best = 1 << 30;
                for (int l = 0; l < SOLVE_LANES; ++l) {
                    const int p_ = S.lane_cnt[l];
                    int r_ = lane_round[l];
                    for (int q = 0; q < 2; ++q) {
                        const int x = q ? m.bB : m.bA;
                        if (x < 0 || b_lane[x] < 0 || b_lane[x] == l) continue;
                        const int need = b_pos[x] < p_ ? b_round[x] : b_round[x] + 1;
                        if (need > r_) r_ = need;
                    }
                    const int key = r_ * 256 + p_;
                    if (key < best) { best = key; ln = l; pos = p_; rd = r_; }
                }
                S.lane_list[ln][pos] = (uint8_t)mi; S.m_round[mi] = (uint8_t)rd; S.lane_cnt[ln] = (uint8_t)(pos + 1);
                lane_round[ln] = (int8_t)rd;
                for (int q = 0; q < 2; ++q) { const int x = q ? m.bB : m.bA; if (x >= 0) { b_lane[x] = (int8_t)ln; b_pos[x] = (int8_t)pos; b_round[x] = (int8_t)rd; } }
                if (rd > max_round) max_round = rd;
                if (pos + 1 > max_cnt) max_cnt = pos + 1;
                const int other = m.bB == b ? m.bA : m.bB;
                if (other < 0) continue;                // static terrain: islands do not propagate across static bodies
                if (flag.test(other)) continue;
                stack[sp++] = other;
                flag.set(other);
            }
            if (b >= 1) {   // joint edges, newest first: hull [hip1, hip0]; upper leg [knee, hip]; lower leg [knee]
                const int w = (b - 1) / 5, r = (b - 1) % 5;
                int jl[2], nj;
                if (r == 0) { jl[0] = 4 * w + 2; jl[1] = 4 * w; nj = 2; }
                else if (r == 1 || r == 3) { jl[0] = 4 * w + (r - 1) + 1; jl[1] = 4 * w + (r - 1); nj = 2; }
                else { jl[0] = 4 * w + (r - 2) + 1; nj = 1; }
                for (int q = 0; q < nj; ++q) {
                    const int ji = jl[q];
                    if (jflag.test(ji)) continue;
                    jflag.set(ji);
                    const int jA = M.jd[ji].bA, jB = M.jd[ji].bB;
                    S.j_island[ji] = (int8_t)isl;
                    S.jorder[w][S.jn[w]++] = (uint8_t)ji;   // a walker's joints only share bodies with each other: its lane runs them in island order
                    const int other = jA == b ? jB : jA;
                    if (flag.test(other)) continue;
                    stack[sp++] = other;
                    flag.set(other);
                }
            }
        }
    }
    S.n_isl = (int8_t)n_isl; S.n_rounds = (int8_t)(max_round + 1); S.max_cnt = (int8_t)max_cnt;
    S.in_island = flag;
}

// ---------------------------------------------------------------- island solver (b2Island::Solve)
// inverse mass, inverse inertia and local centre of the two bodies of a constraint; A = static terrain: zeros
struct MassAB { float mA, iA, mB, iB; V2 lcA, lcB; };
MW_HD_INLINE MassAB mass_of_pair(const Scratch &S, int bA, int bB) {
    MassAB q;
    if (bA < 0) { q.mA = 0.0f; q.iA = 0.0f; q.lcA = v2(0, 0); }
    else { const int sa = shape_of_body(bA); q.mA = S.sh_im[sa]; q.iA = S.sh_ii[sa]; q.lcA = S.sh_lc[sa]; }
    const int sb = shape_of_body(bB);
    q.mB = S.sh_im[sb]; q.iB = S.sh_ii[sb]; q.lcB = S.sh_lc[sb];
    return q;
}

// k = [ex.x ex.y ex.z ey.x ey.y ey.z ez.x ez.y ez.z]; Cramer's rule as b2Mat33::Solve33, split into the part that only
// depends on the matrix (constant over the sweeps of a step) and the part that depends on the right-hand side
MW_HD_INLINE void solve33_prepare(const float *k, float &cx, float &cy, float &cz, float &det) {
    const float exx = k[0], exy = k[1], exz = k[2], eyx = k[3], eyy = k[4], eyz = k[5], ezx = k[6], ezy = k[7], ezz = k[8];
    cx = eyy * ezz - eyz * ezy; cy = eyz * ezx - eyx * ezz; cz = eyx * ezy - eyy * ezx;  // cross(ey, ez)
    det = exx * cx + exy * cy + exz * cz;
    if (det != 0.0f) det = 1.0f / det;
}
MW_HD_INLINE void solve33(const float *k, float cx, float cy, float cz, float det, float bx, float by, float bz, float &x, float &y, float &z) {
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
MW_HD_INLINE void contact_init_warm(Hot &Wd, Manifold &m, const MassAB &q) {
    MW_FLOPS(m.npts == 2 ? 230 : 120);   // two transforms (sin / cos), world manifold, rA / rB, normal and tangent masses, block K and inverse, warm start
    const float mA = q.mA, iA = q.iA, mB = q.mB, iB = q.iB;
    const V2 cA = m.bA < 0 ? v2(0, 0) : Wd.b[m.bA].c, cB = Wd.b[m.bB].c;
    Xf xfA; if (m.bA < 0) { xfA.p = v2(0, 0); xfA.q.s = 0; xfA.q.c = 1; } else xfA = xf_from(Wd.b[m.bA].c, Wd.b[m.bA].a, q.lcA);
    const Xf xfB = xf_from(Wd.b[m.bB].c, Wd.b[m.bB].a, q.lcB);
    V2 normal, pts[2];  // b2WorldManifold::Initialize
    if ((m.type & 1) == 0) {
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
MW_HD_INLINE void joint_init_warm(const Model &M, Hot &Wd, const ColdView &Cd, Scratch &S, int ji, float h, JointCache &c) {
    const JointDef &jd = M.jd[ji];
    const Joint &j = Cd.j[ji];
    MW_FLOPS(140);   // two sin / cos pairs, rA / rB, the 3x3 K, its Cramer terms, warm start
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
// Box2D branches on the limit state (point-to-point only | limit + point, with or without the "reduce" fallback to the 2 x 2 system).
// The lanes of a wavefront hold joints in every one of those states, so a branching version runs all the paths one after the other
// behind exec masks.  Here the paths share what they have in common -- Cdot1, ONE 2 x 2 solve whose right-hand side is selected, one
// application of the impulse -- and the rest is value selects: every lane computes exactly the expressions its own branch of Box2D
// would (a select never changes a value), at two thirds of the instructions and without the exec bookkeeping.
MW_HD_INLINE void joint_solve_velocity(Hot &Wd, JointCache &c) {
    Body &A = Wd.b[c.bA], &B = Wd.b[c.bB];
    const float mA = c.mA, iA = c.iA, mB = c.mB, iB = c.iB;
    const V2 rA = c.rA, rB = c.rB;
    V2 vA = A.v, vB = B.v; float wA = A.w, wB = B.w;
    const int ls = c.limit_state;
    MW_FLOPS((ls != 3 ? 9 : 0) + (ls != 0 ? 73 : 38));
    if (ls != 3) {  // motor (enableMotor is always true)
        const float Cdot = wB - wA - c.motor_speed;
        float imp = -c.motor_mass * Cdot;
        const float old = c.motor_impulse, maxi = c.maxi;
        c.motor_impulse = clampf(old + imp, -maxi, maxi);
        imp = c.motor_impulse - old;
        wA -= iA * imp; wB += iB * imp;
    }
    const bool p2p = ls == 0;   // point-to-point only
    const V2 Cdot1 = vB + cross(wB, rB) - vA - cross(wA, rA);
    const float Cdot2 = wB - wA;
    float jx, jy, jz;           // limit + point constraint (3 x 3); unused (and possibly not finite) on a point-to-point lane
    solve33(c.k, c.c33x, c.c33y, c.c33z, c.idet33, Cdot1.x, Cdot1.y, Cdot2, jx, jy, jz);
    jx = -jx; jy = -jy; jz = -jz;
    const float sum = c.iz + jz;
    const bool reduce = (ls == 1 && sum < 0.0f) || (ls == 2 && sum > 0.0f);
    // the 2 x 2 system: right-hand side -Cdot (point-to-point) or -Cdot1 + m_impulse.z * (ez.x, ez.y) ("reduce")
    const float rx0 = -Cdot1.x, ry0 = -Cdot1.y;
    const float rx1 = rx0 + c.iz * c.k[6], ry1 = ry0 + c.iz * c.k[7];
    float qx, qy;
    solve22(c.k, c.idet22, reduce ? rx1 : rx0, reduce ? ry1 : ry0, qx, qy);
    const bool two = p2p || reduce;
    const float ix = two ? qx : jx, iy = two ? qy : jy;
    const float iz = reduce ? -c.iz : jz;                    // (not applied on a point-to-point lane)
    c.ix += ix; c.iy += iy;
    c.iz = reduce ? 0.0f : (p2p ? c.iz : sum);
    const V2 P = v2(ix, iy);
    const float tA = cross(rA, P), tB = cross(rB, P);
    vA = vA - mA * P; wA -= iA * (p2p ? tA : tA + iz);
    vB = vB + mB * P; wB += iB * (p2p ? tB : tB + iz);
    A.v = vA; A.w = wA; B.v = vB; B.w = wB;
}

// b2ContactSolver::SolveVelocityConstraints for manifold k
// ... on velocities the caller holds (the continuous pass keeps its one moving body in registers over all sweeps)
// `changed` is set when an accumulated impulse ends the call with another value than it began with
// SA: body A is the static terrain (zero velocity, zero inverse mass).  Its terms -- subtracting a zero velocity, adding zero times an
// impulse -- change no value (at most the sign of a zero, which nothing here divides by), so they are left out: the sub-steps of the
// continuous pass, whose contacts are all of that kind, run a third fewer instructions per sweep.
template <bool SA>
MW_HD_INLINE void contact_solve_velocity_t(Manifold &m, const MassAB &q, V2 &vA, float &wA, V2 &vB, float &wB, bool &changed) {
    const float mA = q.mA, iA = q.iA, mB = q.mB, iB = q.iB;
    const float o_n0 = m.ni[0], o_n1 = m.ni[1], o_t0 = m.ti[0], o_t1 = m.ti[1];
    MW_FLOPS(SA ? (m.npts == 2 ? 104 : 50) : (m.npts == 2 ? 152 : 73));
    const V2 normal = m.normal, tangent = cross(normal, 1.0f);
    MW_UNROLL
    for (int i = 0; i < 2; ++i) if (i < m.npts) {  // friction first
        const V2 dv = SA ? vB + cross(wB, m.rB[i]) : vB + cross(wB, m.rB[i]) - vA - cross(wA, m.rA[i]);
        const float vt = dot(dv, tangent);
        float lambda = m.tm[i] * (-vt);
        const float maxf = m.friction * m.ni[i];
        const float newi = clampf(m.ti[i] + lambda, -maxf, maxf);
        lambda = newi - m.ti[i];
        m.ti[i] = newi;
        const V2 P = lambda * tangent;
        if (!SA) { vA = vA - mA * P; wA -= iA * cross(m.rA[i], P); }
        vB = vB + mB * P; wB += iB * cross(m.rB[i], P);
    }
    if (m.npts == 1 || !m.block) {
        MW_UNROLL
    for (int i = 0; i < 2; ++i) if (i < m.npts) {
            const V2 dv = SA ? vB + cross(wB, m.rB[i]) : vB + cross(wB, m.rB[i]) - vA - cross(wA, m.rA[i]);
            const float vn = dot(dv, normal);
            float lambda = -m.nm[i] * (vn - 0.0f);  // restitution 0 -> velocityBias 0
            const float newi = mxf(m.ni[i] + lambda, 0.0f);
            lambda = newi - m.ni[i];
            m.ni[i] = newi;
            const V2 P = lambda * normal;
            if (!SA) { vA = vA - mA * P; wA -= iA * cross(m.rA[i], P); }
            vB = vB + mB * P; wB += iB * cross(m.rB[i], P);
        }
    } else {  // block solver
        const float a1 = m.ni[0], a2 = m.ni[1];
        const V2 dv1 = SA ? vB + cross(wB, m.rB[0]) : vB + cross(wB, m.rB[0]) - vA - cross(wA, m.rA[0]);
        const V2 dv2 = SA ? vB + cross(wB, m.rB[1]) : vB + cross(wB, m.rB[1]) - vA - cross(wA, m.rA[1]);
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
            if (!SA) { vA = vA - mA * (P1 + P2); wA -= iA * (cross(m.rA[0], P1) + cross(m.rA[1], P2)); }
            vB = vB + mB * (P1 + P2); wB += iB * (cross(m.rB[0], P1) + cross(m.rB[1], P2));
            m.ni[0] = x1; m.ni[1] = x2;
        }
    }
    changed = changed || m.ni[0] != o_n0 || m.ni[1] != o_n1 || m.ti[0] != o_t0 || m.ti[1] != o_t1;
}
MW_HD_INLINE void contact_solve_velocity_on(Manifold &m, const MassAB &q, V2 &vA, float &wA, V2 &vB, float &wB, bool &changed) {
    contact_solve_velocity_t<false>(m, q, vA, wA, vB, wB, changed);
}
MW_HD_INLINE void contact_solve_velocity_on(Manifold &m, const MassAB &q, V2 &vA, float &wA, V2 &vB, float &wB) {
    bool changed = false;
    contact_solve_velocity_on(m, q, vA, wA, vB, wB, changed);
}
MW_HD_INLINE void contact_solve_velocity(Hot &Wd, Manifold &m, const MassAB &q) {
    V2 vA = m.bA < 0 ? v2(0, 0) : Wd.b[m.bA].v, vB = Wd.b[m.bB].v;
    float wA = m.bA < 0 ? 0.0f : Wd.b[m.bA].w, wB = Wd.b[m.bB].w;
    contact_solve_velocity_on(m, q, vA, wA, vB, wB);
    if (m.bA >= 0) { Wd.b[m.bA].v = vA; Wd.b[m.bA].w = wA; }
    Wd.b[m.bB].v = vB; Wd.b[m.bB].w = wB;
}

// b2ContactSolver::SolvePositionConstraints for manifold k; returns its minimum separation
MW_HD_INLINE float contact_solve_position(Hot &Wd, const Manifold &m, const MassAB &q) {
    float min_sep = 0.0f;
    MW_FLOPS(137 * (m.type >> 1));   // per point: two transforms (sin / cos), separation, K, impulse
    const float mA = q.mA, iA = q.iA, mB = q.mB, iB = q.iB;
    V2 cA = m.bA < 0 ? v2(0, 0) : Wd.b[m.bA].c, cB = Wd.b[m.bB].c;
    float aA = m.bA < 0 ? 0.0f : Wd.b[m.bA].a, aB = Wd.b[m.bB].a;
    const V2 lcA = q.lcA, lcB = q.lcB;
    const int np = m.type >> 1;   // b2ContactPositionConstraint::pointCount = the manifold's
    // body A static (a terrain edge): its pose is the identity and stays it -- angle +0 (0 - 0 * x), whose sin / cos polynomial gives
    // exactly (0, 1) -- so the polynomial is skipped for it and the same products and sums run on (0, 1)
    const bool static_a = m.bA < 0;
    MW_UNROLL
    for (int i = 0; i < 2; ++i) if (i < np) {
        Xf xfA;
        if (static_a) { xfA.q.s = 0.0f; xfA.q.c = 1.0f; xfA.p = cA - mul(xfA.q, lcA); } else xfA = xf_from(cA, aA, lcA);
        const Xf xfB = xf_from(cB, aB, lcB);
        V2 normal, point; float sep;
        if ((m.type & 1) == 0) {
            normal = mul(xfA.q, m.local_normal);
            const V2 plane = mul(xfA, m.local_point), clip = mul(xfB, m.lp[i]);
            sep = (dot(clip - plane, normal) - POLY_RADIUS) - POLY_RADIUS; point = clip;  // - pc->radiusA - pc->radiusB
        } else {
            normal = mul(xfB.q, m.local_normal);
            const V2 plane = mul(xfB, m.local_point), clip = mul(xfA, m.lp[i]);
            sep = (dot(clip - plane, normal) - POLY_RADIUS) - POLY_RADIUS; point = clip;  // - pc->radiusA - pc->radiusB
            normal = -normal;
        }
        const V2 rA = point - cA, rB = point - cB;
        min_sep = mnf(min_sep, sep);
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
MW_HD_INLINE bool joint_solve_position(Hot &Wd, const JointCache &c) {
    MW_FLOPS(c.limit_state != 0 ? 145 : 135);
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
// After the discrete solve Box2D looks, for every contact between an awake dynamic body and a static one (here: terrain edges),
// for the first time in this step at which the two shapes come within linearSlop of each other (multiwalker_toi.hpp), takes the
// earliest such event, moves the body back to that time, solves a sub-step for the remaining time on a mini island (that body and
// its touching static contacts: 20 TOI position iterations at Baumgarte 0.75, the step's velocity iterations without warm starting,
// integration) and repeats until no event is left (at most 8 sub-steps per contact).  Joints take no part (b2Island::SolveTOI
// ignores them); two dynamic non-bullet bodies are never tested against each other.
MW_HD void poly_aabb_at(const Shape &s, V2 c, float a, float &xmin, float &xmax, float &ymin, float &ymax) {
    const Xf t = xf_from(c, a, s.centroid);
    for (int i = 0; i < s.n; ++i) {
        const V2 p = mul(t, s.v[i]);
        xmin = mnf(xmin, p.x); xmax = mxf(xmax, p.x); ymin = mnf(ymin, p.y); ymax = mxf(ymax, p.y);
    }
}
MW_HD void proxy_of_shape(Proxy &p, const Shape &s) {
    p.n = s.n;
    p.v.e0 = s.v[0]; p.v.e1 = 1 < s.n ? s.v[1] : s.v[0]; p.v.e2 = 2 < s.n ? s.v[2] : s.v[0]; p.v.e3 = 3 < s.n ? s.v[3] : s.v[0]; p.v.e4 = 4 < s.n ? s.v[4] : s.v[0];
}
MW_HD Sweep sweep_of_body(const Model &M, const Hot &Wd, const ColdView &Cd, int b) {
    Sweep s;
    s.lc = M.shape[shape_of_body(b)].centroid;
    s.c0 = Cd.sweep_c0[b]; s.a0 = Cd.sweep_a0[b]; s.alpha0 = Cd.sweep_alpha0[b];
    s.c = Wd.b[b].c; s.a = Wd.b[b].a;
    return s;
}
// the box a body's vertices stay in while it goes from (c0, a0) to (c, a): used to cull time-of-impact computations that cannot
// report an event (never changes a result)
struct SweptBox { float xmin, xmax, ymin, ymax; };
MW_HD SweptBox swept_box(const Shape &sh, const Sweep &sB) {
    MW_FLOPS(170);
    SweptBox q; q.xmin = 3.0e38f; q.xmax = -3.0e38f; q.ymin = 3.0e38f; q.ymax = -3.0e38f;
    poly_aabb_at(sh, sB.c0, sB.a0, q.xmin, q.xmax, q.ymin, q.ymax);
    poly_aabb_at(sh, sB.c, sB.a, q.xmin, q.xmax, q.ymin, q.ymax);
    float r2 = 0.0f;   // a vertex leaves the box of its two end poses by at most |r| (1 - cos(da / 2)) <= |r| da^2 / 8 in between
    for (int i = 0; i < sh.n; ++i) { const V2 r = sh.v[i] - sh.centroid; r2 = mxf(r2, dot(r, r)); }
    const float da = sB.a - sB.a0, mrg = sqrtf(r2) * da * da * 0.125f + LINEAR_SLOP;
    q.xmin -= mrg; q.xmax += mrg; q.ymin -= mrg; q.ymax += mrg;
    return q;
}
// time of impact of body `bi` with terrain edge e, as b2World::SolveTOI computes it for one contact
MW_HD float toi_alpha_terrain(const Model &M, const ColdView &Cd, int bi, int e, const Sweep &sB, const SweptBox &box) {
    const V2 p1 = v2(M.tx[e], Cd.ty[e]), p2 = v2(M.tx[e + 1], Cd.ty[e + 1]);
    const float m = 4.0f * LINEAR_SLOP;   // what the root finder calls touching, with margin
    if (box.xmin - m > p2.x || box.xmax + m < p1.x || box.ymin - m > mxf(p1.y, p2.y) || box.ymax + m < mnf(p1.y, p2.y)) { MW_STAT(toi_culled, 1); return 1.0f; }
    MW_STAT(toi_full, 1); MW_STAT(lane_cost[g_stats_lane()], 6000);
    Proxy pA, pB;
    pA.n = 2; pA.v.e0 = p1; pA.v.e1 = p2; pA.v.e2 = p1; pA.v.e3 = p1; pA.v.e4 = p1;
    proxy_of_shape(pB, M.shape[shape_of_body(bi)]);
    Sweep sA;
    sA.lc = v2(0, 0); sA.c0 = v2(0, 0); sA.c = v2(0, 0); sA.a0 = 0.0f; sA.a = 0.0f; sA.alpha0 = 0.0f;
    float beta;
    const int state = time_of_impact(beta, pA, sA, pB, sB);
    const float alpha0 = sB.alpha0;
    return state == TOI_TOUCHING ? mnf(alpha0 + (1.0f - alpha0) * beta, 1.0f) : 1.0f;
}
// b2ContactSolver::SolveTOIPositionConstraints for one manifold: only the TOI body (B; A is static) moves
MW_HD float contact_solve_toi_position(Hot &Wd, const Manifold &m, const MassAB &q) {
    float min_sep = 0.0f;
    MW_FLOPS(95 * (m.type >> 1));
    const float mB = q.mB, iB = q.iB;
    V2 cB = Wd.b[m.bB].c;
    float aB = Wd.b[m.bB].a;
    Xf xfA; xfA.p = v2(0, 0); xfA.q.s = 0.0f; xfA.q.c = 1.0f;
    const int np = m.type >> 1;
    MW_UNROLL
    for (int i = 0; i < 2; ++i) if (i < np) {
        const Xf xfB = xf_from(cB, aB, q.lcB);
        V2 normal, point; float sep;
        if ((m.type & 1) == 0) {
            normal = mul(xfA.q, m.local_normal);
            const V2 plane = mul(xfA, m.local_point), clip = mul(xfB, m.lp[i]);
            sep = (dot(clip - plane, normal) - POLY_RADIUS) - POLY_RADIUS; point = clip;  // - pc->radiusA - pc->radiusB
        } else {
            normal = mul(xfB.q, m.local_normal);
            const V2 plane = mul(xfB, m.local_point), clip = mul(xfA, m.lp[i]);
            sep = (dot(clip - plane, normal) - POLY_RADIUS) - POLY_RADIUS; point = clip;  // - pc->radiusA - pc->radiusB
            normal = -normal;
        }
        const V2 rB = point - cB;
        min_sep = mnf(min_sep, sep);
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

// the next terrain contact of body b after the one with key `below` in its contact-edge list (descending key), touching or not
MW_HD int next_terrain_slot(const Model &M, const ColdView &Cd, int b, SlotBits occ, uint64_t below, uint64_t &key_out) {
    int best = -1;
    uint64_t bk = 0;
    const int base = M.slot_base[b];
    while (occ.any()) {   // occ: occupied_slots of the body's cache
        const int k = occ.pop_lowest();
        const Slot &sl = Cd.slot[base + k];
        const uint64_t key = b == 0 ? contact_key(sl.batch, 0, proxy_of_edge(sl.edge)) : contact_key(sl.batch, proxy_of_edge(sl.edge), proxy_of_body(b, M.NT));
        if (key < below && (best < 0 || key > bk)) { best = base + k; bk = key; }
    }
    key_out = bk;
    return best;
}
// b2Contact::Update inside the continuous pass: events take effect at once, in call order (ContactDetector, :50-84)
MW_HD bool toi_update_contact(const Model &M, Hot &Wd, const ColdView &Cd, Slot &sl, int b, ManifoldOut &mo) {
    mo.npts = 0;
    edge_polygon_manifold(M, Cd, sl.edge, M.shape[shape_of_body(b)], body_xf(M, Wd.b[b], b), mo);
    const int ev = contact_update(sl, mo);
    if (ev == 1) contact_begin_flags(Wd, -1, b);
    if (ev != 0 && is_lower_leg(b)) set_ground_flag(Wd, b, ev == 1);
    return sl.touching != 0;
}

// b2World::SolveTOI.  Box2D's loop -- find the contact with the smallest time of impact over the whole world, handle it, repeat -- only
// ever tests a dynamic body against the static terrain here (two dynamic non-bullet bodies are skipped), and handling an event changes
// nothing but that one body, its own contacts and their cached times.  So the events of one body form a chain that does not depend on
// the other bodies' chains, and the chains run side by side, one body per lane at a time.  What the world-wide ORDER of the events
// decides is only the number of the FindNewContacts call that follows each of them (contact_key: the place of the contacts it creates
// in Box2D's lists) and which box of the other body a new package / hull pair is tested against.  Every chain therefore logs its events
// (time, contact), numbers the contacts it creates provisionally, and afterwards one lane merges the logs into Box2D's order -- smallest
// time first, among equal times the contact nearest the front of the world's list -- hands out the final numbers and creates the pairs.
// events of one env in one SolveTOI that the log holds (sticky Hot::overflow bit 2 past it; seen: <= 7 with up to four walkers, and at most
// two per walker beyond -- 10, 12, 14, 16, 18, 20 for 5 .. 10 walkers, in the step that follows a reset, when both lower legs of every
// walker arrive at the ground at once: 256 envs x 200 steps each)
constexpr int TOI_MAX_EVENTS = MAX_WALKERS <= 4 ? 16 : 4 * MAX_WALKERS;
constexpr int TOI_MAX_PAIR_EVENTS = MAX_WALKERS <= 4 ? 4 : MAX_WALKERS + 2;   // ... those of them that moved the proxy of the package or of a hull
struct ToiEvent { float alpha; uint16_t slot, batch; uint8_t body, idx, moved, fat_i; };
struct ToiWork {            // shared by the lanes of an env (LDS in the HIP kernel)
    int n_ev, n_fat;
    uint32_t batch_base;    // Hot::batch when the pass began
    uint32_t overflow;      // bits for Hot::overflow, gathered by the lanes
    ToiEvent ev[TOI_MAX_EVENTS];
    float fat_log[TOI_MAX_PAIR_EVENTS][4];       // the fat AABB after an event that moved the proxy of the package or a hull
    float fat0[1 + MAX_WALKERS][4];              // ... and their boxes when the pass began (package, hull 0, hull 1, ...)
    // the bodies whose first search found an event (see solve_toi): body, the event's contact slot (index into the body's cache), its time
    int n_pend;
    struct { uint8_t body, k; uint16_t pad_; float alpha; } pend[MAXB];
#if MW_CAPW > 4
    // the merge's bookkeeping (one lane; a local array up to four walkers, see Scratch::bi_lane).  (Its other array, the current boxes of the
    // package and the hulls, is fat0 itself: nothing reads the boxes of the pass's beginning once the merge has started.)
    uint8_t mg_next_idx[MAXB];
#endif
};
struct ToiLaneWork {        // per lane: the cached times of impact of the contacts of the body it is working on
    float *alpha;           // [Model::slot_cap of the body]
    uint8_t *meta;          // bit 0 e_toiFlag (the cached time is valid), bit 1 NOT e_enabledFlag, bits 2.. m_toiCount
    Manifold *ovf;          // room for the mini island's manifolds past the lane-private ones
    int ovf_cap;
};
constexpr int TOI_MREG = 4;           // manifolds of a mini island held in lane-private storage (registers in the HIP kernel)
MW_HD int pair_body_index(int b) { return b == 0 ? 0 : 1 + (b - 1) / 5; }   // package / hull -> row of ToiWork::fat0

// the final number of the FindNewContacts call after event `idx` of body b's chain (valid once the merge has reached it)
MW_HD uint32_t toi_final_batch(const ToiWork &T, int n, int b, int idx) {
    for (int i = 0; i < n; ++i) if (T.ev[i].body == b && T.ev[i].idx == idx) return T.batch_base + 1u + T.ev[i].batch;
    return T.batch_base;
}

// one body's chain of events
#ifdef MW_STATS
inline int g_stats_lane() { return (int)(g_stats.cur_lane & 3); }
#endif
// first_k == TOI_SEARCH_ONLY: the chain's first search alone -- an event it finds is put on ToiWork::pend and the chain stops there;
// first_k >= 0: the chain from that event on (the search that found it is not repeated: slot first_k of the body's cache at time first_alpha)
constexpr int TOI_SEARCH_ONLY = -2;
template <class Par>
MW_HD void toi_body_chain(const Model &M, Hot &Wd, const ColdView &Cd, const Scratch &S, ToiWork &T, ToiLaneWork &TL, Par par, int mover, float h, int first_k, float first_alpha) {
    const int base = M.slot_base[mover], cap = M.slot_cap[mover];
    const Shape &msh = M.shape[shape_of_body(mover)];
#ifdef MW_STATS
    g_stats.cur_lane = mover;
#endif
    Cd.sweep_alpha0[mover] = 0.0f;   // "if (m_stepComplete)": alpha0 = 0, every contact's cached TOI invalid, its sub-step count 0, enabled
    if (!Wd.awake.test(mover)) return;   // a sleeping body against static terrain: no active body
    SlotBits occ = occupied_slots(Cd.slot + base, cap);
    if (!occ.any()) return;
    for (int k = 0; k < cap; ++k) TL.meta[k] = 0;
    const float fr = sqrtf(FRICTION * msh.friction);
    const MassAB qm = mass_of_pair(S, -1, mover);
    const int pb = proxy_of_body(mover, M.NT);
    SweptBox box = swept_box(msh, sweep_of_body(M, Wd, Cd, mover));
    int n_events = 0;   // events of this chain that reached FindNewContacts
    for (int guard = 0; guard < (MAX_SUB_STEPS + 2) * cap; ++guard) {
        // ---- this body's contact with the smallest time of impact; among equal times the first of the contact list (largest key)
        MW_TACC_T0(ta_);
        float min_alpha = 1.0f;
        uint64_t min_key = 0;
        int min_k = -1;
        if (guard == 0 && first_k >= 0) { min_k = first_k; min_alpha = first_alpha; }   // found by the search-only call (nothing has touched the body since)
        else
        for (SlotBits o2 = occ; o2.any();) {
            const int k = o2.pop_lowest();
            const Slot &sl = Cd.slot[base + k];
            const uint8_t meta = TL.meta[k];
            if ((meta & 2) || (meta >> 2) > MAX_SUB_STEPS) continue;
            if (!(meta & 1)) {   // no valid cached TOI: compute it on the body's current sweep
                TL.alpha[k] = toi_alpha_terrain(M, Cd, mover, sl.edge, sweep_of_body(M, Wd, Cd, mover), box);
                TL.meta[k] = (uint8_t)(meta | 1);
            }
            const float alpha = TL.alpha[k];
            if (alpha > min_alpha) continue;
            const uint64_t key = mover == 0 ? contact_key(sl.batch, 0, proxy_of_edge(sl.edge)) : contact_key(sl.batch, proxy_of_edge(sl.edge), pb);
            if (alpha < min_alpha || (min_k >= 0 && key > min_key)) { min_alpha = alpha; min_k = k; min_key = key; }
        }
        MW_TACC(first_k == TOI_SEARCH_ONLY ? 6 : 0, ta_);
        if (min_k < 0 || 1.0f - 10.0f * B2_EPSILON < min_alpha || M.continuous == 2) break;   // no more TOI events
        if (first_k == TOI_SEARCH_ONLY) {
            const int pi = mover == 0 ? 0 : par.alloc(&T.n_pend);   // entry 0 is the package's (see solve_toi); at most one entry per body: MAXB is room enough
            T.pend[pi].body = (uint8_t)mover; T.pend[pi].k = (uint8_t)min_k; T.pend[pi].pad_ = 0; T.pend[pi].alpha = min_alpha;
            break;
        }
        const int min_slot = base + min_k;
        MW_PHASE(4);
        // ---- advance the body to the time of impact (b2Body::Advance); the static edge does not move
        const V2 bk_c0 = Cd.sweep_c0[mover], bk_c = Wd.b[mover].c;
        const float bk_a0 = Cd.sweep_a0[mover], bk_a = Wd.b[mover].a, bk_alpha0 = Cd.sweep_alpha0[mover];
        {
            Sweep sw = sweep_of_body(M, Wd, Cd, mover);
            sweep_advance(sw, min_alpha);
            Cd.sweep_c0[mover] = sw.c0; Cd.sweep_a0[mover] = sw.a0; Cd.sweep_alpha0[mover] = sw.alpha0;
            Wd.b[mover].c = sw.c0; Wd.b[mover].a = sw.a0;
        }
        ManifoldOut mo;
        const bool touching = toi_update_contact(M, Wd, Cd, Cd.slot[min_slot], mover, mo);   // the TOI contact likely has some new contact points
        TL.meta[min_k] = (uint8_t)((TL.meta[min_k] & ~1u) + 4u);   // e_toiFlag cleared, ++m_toiCount
        MW_STAT(toi_events, 1);
#ifdef MW_STATS
        if (g_stats.ev_n < 64) { g_stats.ev_body[g_stats.ev_n] = (unsigned char)mover; g_stats.ev_sweeps[g_stats.ev_n] = 0; }
        g_stats.ev_n += 1;
#endif
        if (!touching) {  // not solid after all: disable the contact, restore the sweep
            MW_STAT(toi_undone, 1);
            TL.meta[min_k] |= 2;
            Cd.sweep_c0[mover] = bk_c0; Cd.sweep_a0[mover] = bk_a0; Cd.sweep_alpha0[mover] = bk_alpha0;
            Wd.b[mover].c = bk_c; Wd.b[mover].a = bk_a;
            continue;
        }
        MW_TACC(1, ta_);
        // ---- mini island: the event's contact, then the body's other contacts with static bodies in its contact-edge order, each
        // updated at the time-of-impact pose and added when it touches
        Manifold mm[TOI_MREG];
        int n_isl = 0;
        auto add_manifold = [&](const ManifoldOut &o, int slot_index) {
            if (n_isl >= MAX_TOI_CONTACTS) return;
            if (n_isl >= TOI_MREG + TL.ovf_cap) { par.or_bits(&T.overflow, 1u); return; }
            Manifold m;
            m.bA = -1; m.bB = (int8_t)mover; m.slot = (int16_t)slot_index; m.npts = (uint8_t)o.npts; m.type = (uint8_t)(o.type | (o.npts << 1)); m.island = 0; m.block = 0;
            m.local_normal = o.local_normal; m.local_point = o.local_point;
            MW_UNROLL
            for (int i = 0; i < 2; ++i) { m.lp[i].x = i < o.npts ? o.lp[i].x : 0.0f; m.lp[i].y = i < o.npts ? o.lp[i].y : 0.0f; m.ni[i] = 0.0f; m.ti[i] = 0.0f; }  // subStep.warmStarting = false
            m.friction = fr;
            m.normal = v2(0, 0); m.rA[0] = m.rA[1] = m.rB[0] = m.rB[1] = v2(0, 0); m.nm[0] = m.nm[1] = m.tm[0] = m.tm[1] = 0.0f;
            m.k11 = m.k12 = m.k22 = m.im11 = m.im12 = m.im22 = 0.0f;
            MW_UNROLL
            for (int q = 0; q < TOI_MREG; ++q) if (n_isl == q) mm[q] = m;
            if (n_isl >= TOI_MREG) TL.ovf[n_isl - TOI_MREG] = m;
            ++n_isl;
        };
        add_manifold(mo, min_slot);
        {
            uint64_t below = ~0ull, key;
            for (int si = next_terrain_slot(M, Cd, mover, occ, below, key); si >= 0; si = next_terrain_slot(M, Cd, mover, occ, below, key)) {
                below = key;
                if (si == min_slot) continue;
                if (n_isl >= MAX_TOI_CONTACTS) break;
                ManifoldOut o2;
                // b2Contact::Update begins with "Re-enable this contact": a contact that an earlier event of this chain found not solid after
                // all (disabled: bit 1 of its meta byte) is a candidate again once a later event's island has looked at it -- a leg tip
                // on the vertex two edges share: the first edge's event is undone, the second one's is solid, and later in the step the
                // first edge is hit for real (scripts/mw_soak.py --seed 5000 --steps 6000 at ten walkers, env-step 200 532 of that run)
                TL.meta[si - base] &= (uint8_t)~2u;
                if (toi_update_contact(M, Wd, Cd, Cd.slot[si], mover, o2)) add_manifold(o2, si);
            }
        }
        // F_ over every manifold of the island, in its order
#define MW_ISLAND_SWEEP(F_)                                                                                       \
        {                                                                                                         \
            MW_UNROLL                                                                                             \
            for (int q_ = 0; q_ < TOI_MREG; ++q_) if (q_ < n_isl) { Manifold &m_ = mm[q_]; F_; }                   \
            for (int q_ = TOI_MREG; q_ < n_isl; ++q_) { Manifold &m_ = TL.ovf[q_ - TOI_MREG]; F_; }                \
        }
        MW_TACC(2, ta_);
        // ---- b2Island::SolveTOI
        for (int it = 0; it < 20; ++it) {   // subStep.positionIterations = 20
            float ms_min = 0.0f;
            MW_ISLAND_SWEEP(ms_min = mnf(ms_min, contact_solve_toi_position(Wd, m_, qm)))
            if (ms_min >= -1.5f * LINEAR_SLOP) break;
        }
        MW_TACC(3, ta_);
        Cd.sweep_c0[mover] = Wd.b[mover].c; Cd.sweep_a0[mover] = Wd.b[mover].a;  // "leap of faith to new safe state"
        {   // InitializeVelocityConstraints on velocities that stay untouched (impulses are zero: nothing to warm start)
            const V2 v_keep = Wd.b[mover].v; const float w_keep = Wd.b[mover].w;
            MW_ISLAND_SWEEP(contact_init_warm(Wd, m_, qm))
            Wd.b[mover].v = v_keep; Wd.b[mover].w = w_keep;
        }
        {
            // Box2D runs all the step's velocity iterations; once a whole sweep leaves every accumulated impulse and the body's
            // velocity exactly unchanged, every further sweep is the same no-op, so stopping there changes no bit of the result.
            // The moving body's velocity stays in registers over the sweeps (the static side never changes).
            V2 vB = Wd.b[mover].v, vA = v2(0, 0);
            float wB = Wd.b[mover].w, wA = 0.0f;
            // The sweep is a deterministic map of (velocity, accumulated impulses).  Besides the exact fixed point it often ends in a
            // cycle (impulses flipping in their last bits, with a period of 2 .. a few dozen sweeps): once the state equals, bit for bit,
            // the state saved after an earlier sweep -- saved at sweeps 1, 2, 4, 8, ... (Brent) -- everything from here on repeats with
            // that period, and the state after the last of the VEL_ITERS sweeps is reached by running only the remainder.
            constexpr int NST = 3 + 4 * TOI_MREG;
            uint32_t anchor[NST];
            int anchor_it = -1, next_anchor = 1;
            const bool track = n_isl <= TOI_MREG;
            auto state_bits = [&](int q) -> uint32_t {   // q: compile-time after unrolling
                float f = q == 0 ? vB.x : (q == 1 ? vB.y : (q == 2 ? wB : 0.0f));
                MW_UNROLL
                for (int k = 0; k < TOI_MREG; ++k) {
                    if (q == 3 + 4 * k) f = mm[k].ni[0];
                    if (q == 4 + 4 * k) f = mm[k].ni[1];
                    if (q == 5 + 4 * k) f = mm[k].ti[0];
                    if (q == 6 + 4 * k) f = mm[k].ti[1];
                }
                uint32_t u; memcpy(&u, &f, 4); return u;
            };
            int stop_at = VEL_ITERS;   // sweeps [0, stop_at) are run
            for (int it = 0; it < stop_at; ++it) {
                MW_STAT(toi_vel_iters, 1); MW_STAT(lane_cost[mover & 3], 300 * n_isl + 60);
#ifdef MW_STATS
                if (g_stats.ev_n > 0 && g_stats.ev_n <= 64) g_stats.ev_sweeps[g_stats.ev_n - 1] += 1;
#endif
                bool changed = false;
                MW_ISLAND_SWEEP(contact_solve_velocity_t<true>(m_, qm, vA, wA, vB, wB, changed))
                if (!changed) { MW_STAT(toi_hist[it / 20], 1); break; }   // no impulse moved: the velocity did not either, and every further sweep is this one
                if (!track || stop_at != VEL_ITERS) { if (it == VEL_ITERS - 1) MW_STAT(toi_hist[9], 1); continue; }
                if (anchor_it >= 0) {
                    bool same = true;
                    MW_UNROLL
                    for (int q = 0; q < NST; ++q) if (q < 3 + 4 * n_isl) same = same && state_bits(q) == anchor[q];
                    if (same) {   // state(it) == state(anchor_it): period it - anchor_it; the last sweep's state = state(it + left)
                        MW_STAT(toi_hist[8], 1);
                        const int period = it - anchor_it;
                        stop_at = it + 1 + (VEL_ITERS - 1 - it) % period;
                        continue;
                    }
                }
                if (it + 1 == next_anchor) {
                    MW_UNROLL
                    for (int q = 0; q < NST; ++q) anchor[q] = state_bits(q);
                    anchor_it = it; next_anchor *= 2;
                }
                if (it == VEL_ITERS - 1) MW_STAT(toi_hist[9], 1);
            }
            MW_STAT(toi_nisl[n_isl < 5 ? n_isl : 5], 1);
            Wd.b[mover].v = vB; Wd.b[mover].w = wB;
        }
#undef MW_ISLAND_SWEEP
        MW_TACC(4, ta_);
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
        // ---- the displaced body: SynchronizeFixtures, every one of its contacts loses its cached time of impact, FindNewContacts.  The
        // call's number is not known yet: the new contacts carry this chain's running count (enough to order this body's own contact
        // list) and Slot::reserved_ = 1 + the event's index in the chain until the merge below.
        {
            const bool moved = sync_fixture(M, Wd, Cd, mover);
            for (int k = 0; k < cap; ++k) TL.meta[k] &= (uint8_t)~1u;
            const int li = par.alloc(&T.n_ev);
            if (li < TOI_MAX_EVENTS) {
                ToiEvent &e = T.ev[li];
                e.alpha = min_alpha; e.slot = (uint16_t)min_slot; e.batch = 0; e.body = (uint8_t)mover; e.idx = (uint8_t)n_events; e.moved = moved ? 1 : 0; e.fat_i = 255;
                if (moved && (mover == 0 || is_hull(mover))) {
                    const int fi = par.alloc(&T.n_fat);
                    if (fi < TOI_MAX_PAIR_EVENTS) { e.fat_i = (uint8_t)fi; for (int q = 0; q < 4; ++q) T.fat_log[fi][q] = Cd.fat[mover][q]; }
                    else par.or_bits(&T.overflow, 4u);
                }
            } else par.or_bits(&T.overflow, 4u);
            if (moved) {
                find_new_terrain_contacts(M, Wd, Cd, mover, T.batch_base + 1u + (uint32_t)n_events, (uint16_t)(1 + n_events));
                occ = occupied_slots(Cd.slot + base, cap);
            }
            ++n_events;
            box = swept_box(msh, sweep_of_body(M, Wd, Cd, mover));   // its new sweep: from the safe pose to the end of the sub-step
            MW_TACC(5, ta_);
            MW_PHASE(3);
        }
    }
}

template <class Par>
MW_HD void solve_toi(const Model &M, Hot &Wd, const ColdView &Cd, Scratch &S, ToiWork &T, ToiLaneWork &TL, Par par, float h) {
    const int L0 = par.lane(), LN = par.n();
    const int NB = M.NB;
    if (L0 == 0) {
        T.n_ev = 0; T.n_fat = 0; T.overflow = 0; T.batch_base = Wd.batch;
        T.n_pend = 1; T.pend[0].body = 255;   // entry 0 is kept for the package: whichever pass, its chain runs on lane 0, the one whose
                                              // time-of-impact cache has room for the package's contact slots (ToiLaneWork, MwDev::toi_lane0_bytes)
        for (int q = 0; q < 4; ++q) T.fat0[0][q] = Cd.fat[0][q];
        for (int w = 0; w < M.W; ++w) for (int q = 0; q < 4; ++q) T.fat0[1 + w][q] = Cd.fat[hull_of(w)][q];
    }
    par.sync();
    if (M.continuous == 3) return;   // timing experiments only: 3 = set-up, 2 = set-up and one search without events
    MW_TSTAMP(1, 2);
    // ---- the chains.  An event is the expensive part of a chain (contact update, mini island, 20 position iterations, up to 180 velocity
    // sweeps, a new search) and only one body in twenty has one in a step -- but among the 64 bodies that the lanes of a wavefront work on
    // at the same moment there nearly always is one, and all of them wait for it.  So the first search of EVERY body runs first (pass 0: the
    // bodies dealt to the lanes in Model::toi_body order; a body with an event goes on ToiWork::pend), and then the pending chains are
    // dealt to the env's lanes (pass 1): the events of a step run side by side instead of one group of bodies after the other.  A chain
    // touches nothing but its own body and that body's contacts with the static terrain, so when it runs decides nothing.
    for (int pass = 0; pass < 2; ++pass) {
        const int n_items = pass == 0 ? M.n_toi : (T.n_pend < NB ? T.n_pend : NB);
        for (int k = L0; k < n_items; k += LN) {
            const int body = pass == 0 ? M.toi_body[k] : T.pend[k].body;
            if (body == 255) continue;   // (the package has no event)
            toi_body_chain(M, Wd, Cd, S, T, TL, par, body, h, pass == 0 ? TOI_SEARCH_ONLY : (int)T.pend[k].k, pass == 0 ? 0.0f : T.pend[k].alpha);
        }
        par.sync();
        if (pass == 0) { MW_TSTAMP(1, 6); }
    }
    MW_TSTAMP(1, 3);
    const int n = T.n_ev < TOI_MAX_EVENTS ? T.n_ev : TOI_MAX_EVENTS;
    MW_TVAL(1, 0, n);
    if (n == 0 && T.overflow == 0) return;
    // ---- merge: Box2D's order of the events; event number r is followed by FindNewContacts call batch_base + 1 + r
    if (L0 == 0) {
#if MW_CAPW > 4
        uint8_t (&next_idx)[MAXB] = T.mg_next_idx;
        float (&cur_fat)[1 + MAX_WALKERS][4] = T.fat0;
#else
        uint8_t next_idx[MAXB];
        float cur_fat[1 + MAX_WALKERS][4];
        for (int p = 0; p <= M.W; ++p) for (int q = 0; q < 4; ++q) cur_fat[p][q] = T.fat0[p][q];
#endif
        for (int b = 0; b < NB; ++b) next_idx[b] = 0;
        for (int r = 0; r < n; ++r) {
            int best = -1;
            float best_alpha = 0.0f;
            uint64_t best_key = 0;
            for (int i = 0; i < n; ++i) {
                const ToiEvent &e = T.ev[i];
                if (e.idx != next_idx[e.body]) continue;   // not the head of its chain (or already merged)
                const Slot &sl = Cd.slot[e.slot];
                const uint32_t batch = sl.reserved_ ? toi_final_batch(T, n, e.body, sl.reserved_ - 1) : (uint32_t)sl.batch;
                const uint64_t key = e.body == 0 ? contact_key(batch, 0, proxy_of_edge(sl.edge)) : contact_key(batch, proxy_of_edge(sl.edge), proxy_of_body(e.body, M.NT));
                if (best < 0 || e.alpha < best_alpha || (e.alpha == best_alpha && key > best_key)) { best = i; best_alpha = e.alpha; best_key = key; }
            }
            ToiEvent &e = T.ev[best];
#ifdef MW_STATS
            for (int i = 0; i < n; ++i) if (i != best && T.ev[i].idx == next_idx[T.ev[i].body] && T.ev[i].body != e.body && T.ev[i].alpha == e.alpha) { MW_STAT(toi_ties, 1); break; }
            if (r == 0) { bool multi = false; for (int i = 1; i < n; ++i) multi = multi || T.ev[i].body != T.ev[0].body; if (multi) MW_STAT(toi_multi, 1); }
            if (e.body == 0 || is_hull(e.body)) MW_STAT(toi_hullpkg, 1);
#endif
            e.batch = (uint16_t)r;
            next_idx[e.body] += 1;
            if (e.moved && (e.body == 0 || is_hull(e.body))) {   // FindNewContacts for the pairs of the moved proxy, against the boxes as they were then
                if (e.fat_i != 255) for (int q = 0; q < 4; ++q) cur_fat[pair_body_index(e.body)][q] = T.fat_log[e.fat_i][q];
                for (int p = 0; p < M.n_dyn_pairs; ++p) {
                    const int bA = M.dyn_a[p], bB = M.dyn_b[p];
                    if (bA != e.body && bB != e.body) continue;
                    Slot &ps = Cd.slot[M.dyn_slot_base + p];
                    if (ps.edge >= 0) continue;
                    const float *fa = cur_fat[pair_body_index(bA)], *fb = cur_fat[pair_body_index(bB)];
                    AABB A, B; A.lx = fa[0]; A.ly = fa[1]; A.hx = fa[2]; A.hy = fa[3]; B.lx = fb[0]; B.ly = fb[1]; B.hx = fb[2]; B.hy = fb[3];
                    if (!aabb_overlap(A, B)) continue;
                    ps.edge = 0; ps.npts = 0; ps.touching = 0; ps.batch = (uint16_t)(T.batch_base + 1u + (uint32_t)r);
                    for (int q = 0; q < 2; ++q) { const int b = q ? bB : bA; if (!Wd.awake.test(b)) { Wd.awake.set(b); Cd.sleep_time[b] = 0.0f; } }   // AddPair: "Wake up the bodies"
                    MW_STAT(toi_pairs, 1);
                }
            }
        }
        Wd.batch = T.batch_base + (uint32_t)n;
        if (T.overflow) Wd.overflow |= (uint8_t)T.overflow;
    }
    par.sync();
    // ---- the contacts created in this pass get the final number of their FindNewContacts call
    for (int bi = L0; bi < NB; bi += LN) {
        const int base = M.slot_base[bi], cap = M.slot_cap[bi];
        for (int k = 0; k < cap; ++k) {
            Slot &sl = Cd.slot[base + k];
            if (sl.reserved_ == 0) continue;
            if (sl.edge >= 0) sl.batch = (uint16_t)toi_final_batch(T, n, bi, sl.reserved_ - 1);
            sl.reserved_ = 0;
        }
    }
    par.sync();
}

// b2World::Step(1/50, 180, 60) for the lanes of `par`, in three phases -- the HIP build runs them as three kernels (the registers of
// the narrow phase and of the time-of-impact root finder would otherwise be charged to the 180-sweep solver loop):
//   step_collide   b2ContactManager::Collide + the island construction and solver schedule of b2World::Solve
//   step_solve     b2Island::Solve of every island, sleeping;  step_post: SynchronizeFixtures, FindNewContacts
//   solve_toi      b2World::SolveTOI
template <class Par>
MW_HD_INLINE void step_collide(const Model &M, Hot &Wd, const ColdView &Cd, Scratch &S, Manifold *MP, Par par) {
    const int L0 = par.lane(), LN = par.n();
    const int NB = M.NB, NDP = M.n_dyn_pairs;
    for (int sh = L0; sh < N_SHAPES; sh += LN) { S.sh_im[sh] = M.shape[sh].inv_mass; S.sh_ii[sh] = M.shape[sh].inv_I; S.sh_lc[sh] = M.shape[sh].centroid; }
    if (L0 == 0) { S.nm = 0; S.moved = BodyBits::none(); }
    par.sync();
    // ---- b2ContactManager::Collide: every contact is updated (D4); Begin / EndContact -> ContactDetector flags
    for (int bi = L0; bi < NB; bi += LN) collide_body_terrain(M, Wd, Cd, S, MP, par, bi);
    par.sync();
    // package - hull and hull - hull pairs (n (n + 1) / 2 of them: 55 with ten walkers), dealt over the lanes: a pair touches its own cache slot,
    // sets flags that only ever go from 0 to 1 and wakes bodies through an atomic OR, so which lane runs it when decides nothing
    for (int p = L0; p < NDP; p += LN) collide_dyn_pair(M, Wd, Cd, S, MP, par, p);
    par.sync();
    // ---- b2World::Solve: islands, constraint order and schedule (one lane)
    if (L0 == 0) build_islands(M, Wd, Cd, S, MP);
    par.sync();
}

// what SolvePositionConstraints reads of a manifold / a joint, fetched again (see step_solve)
MW_HD_INLINE void manifold_position_part(Manifold &m, const Manifold &src) {
    m.bA = src.bA; m.bB = src.bB; m.type = src.type; m.island = src.island;
    m.local_normal = src.local_normal; m.local_point = src.local_point; m.lp[0] = src.lp[0]; m.lp[1] = src.lp[1];
}
// what SolveVelocityConstraints reads of a manifold (b2ContactVelocityConstraint after InitializeVelocityConstraints), and the four
// accumulated impulses it leaves behind.  PM: a pointer to the working copy -- in the HIP build one that says which memory it points
// into (MW_LDS: were both kinds handed over as plain pointers, hipcc would fold the two fetches into one flat load through a selected
// pointer); scalar by scalar because a struct cannot be copied out of a qualified address space
template <class PM>
MW_HD_INLINE void manifold_velocity_part_from(Manifold &m, PM s) {
    m.bA = s->bA; m.bB = s->bB; m.npts = s->npts; m.block = s->block;
    m.normal.x = s->normal.x; m.normal.y = s->normal.y;
    MW_UNROLL
    for (int i = 0; i < 2; ++i) {
        m.rA[i].x = s->rA[i].x; m.rA[i].y = s->rA[i].y; m.rB[i].x = s->rB[i].x; m.rB[i].y = s->rB[i].y;
        m.nm[i] = s->nm[i]; m.tm[i] = s->tm[i]; m.ni[i] = s->ni[i]; m.ti[i] = s->ti[i];
    }
    m.friction = s->friction;
    m.k11 = s->k11; m.k12 = s->k12; m.k22 = s->k22; m.im11 = s->im11; m.im12 = s->im12; m.im22 = s->im22;
}
template <class PM>
MW_HD_INLINE void manifold_store_impulses_to(PM d, const Manifold &m) {
    d->ni[0] = m.ni[0]; d->ni[1] = m.ni[1]; d->ti[0] = m.ti[0]; d->ti[1] = m.ti[1];
}
template <class PM>
MW_HD_INLINE void manifold_position_part_from(Manifold &m, PM s) {
    m.bA = s->bA; m.bB = s->bB; m.type = s->type; m.island = s->island;
    m.local_normal.x = s->local_normal.x; m.local_normal.y = s->local_normal.y;
    m.local_point.x = s->local_point.x; m.local_point.y = s->local_point.y;
    MW_UNROLL
    for (int i = 0; i < 2; ++i) { m.lp[i].x = s->lp[i].x; m.lp[i].y = s->lp[i].y; }
}
#ifndef MW_LDS
#define MW_LDS   // the qualifier of a pointer into the solver's LDS working copies (multiwalker_impl.hpp: address_space(3)); nothing on the CPU
#endif
MW_HD_INLINE void joint_position_part(const Model &M, const Scratch &S, int ji, JointCache &c) {
    const JointDef &jd = M.jd[ji];
    const MassAB q = mass_of_pair(S, jd.bA, jd.bB);
    c.bA = jd.bA; c.bB = jd.bB;
    c.mA = q.mA; c.iA = q.iA; c.mB = q.mB; c.iB = q.iB;
    c.lA = jd.lA - q.lcA; c.lB = jd.lB - q.lcB;
    c.lower = jd.lower; c.upper = jd.upper;
    float mm = q.iA + q.iB;
    if (mm > 0.0f) mm = 1.0f / mm;
    c.motor_mass = mm;
}

// One lane of the island solver: its walker's joints (in island order) and the first Par::MREG manifolds of its list
template <int NREG>
struct SolveLane {
    JointCache jc[4];
    int ji[4], jisl[4], jn;   // joint ids in island order and their islands
    Manifold mc[NREG];
    MassAB mq[NREG];
    int mrd[NREG], mix[NREG];   // round and pool index of mc[r] (-1: none)
    int cnt;                    // manifolds in the lane's list
    int mo_base;                // MO slot of the lane's first manifold past mc[]
};

// b2Island::Solve of every island of the step at once (islands share nothing, so running them together changes nothing), then the
// sleep test.  MP = the manifold pool of the step (Scratch::m).  The HIP solver launch leaves it in HBM: a lane works on register
// copies of the first Par::MREG manifolds of its list and on copies in MO -- LDS, room for `mo_cap` manifolds per env -- of the rest;
// whatever does not fit there either is solved in place in MP.
template <class Par>
MW_HD_INLINE void step_solve(const Model &M, Hot &Wd, const ColdView &Cd, Scratch &S, Manifold *MP, Manifold *MO, int mo_cap, Par par) {
    const float h = 1.0f / FPS;
    const int L0 = par.lane(), LN = par.n();
    const int NB = M.NB, NW = M.W;   // the model scalars are read once: the solver loops below must not go back to memory for them
    const int n_rounds = S.n_rounds, max_cnt = S.max_cnt;
    // ---- integrate velocities (gravity + the pending initial push) of the bodies of this step's islands
    for (int bi = L0; bi < NB; bi += LN) {
        if (S.island_of[bi] < 0) continue;
        Body &b = Wd.b[bi];
        Cd.sweep_c0[bi] = b.c; Cd.sweep_a0[bi] = b.a;   // b2Island::Solve: "store positions for continuous collision"
        float fx = 0.0f;
        if (is_hull(bi)) fx = Wd.push_x[(bi - 1) / 5];
        const float im = S.sh_im[shape_of_body(bi)];
        b.v.x += h * (GRAVITY_Y * 0.0f + im * fx);
        b.v.y += h * (GRAVITY_Y + im * 0.0f);
        // linear / angular damping are 0: v *= 1 / (1 + h * 0)
    }
    par.sync();
    for (int w = L0; w < NW; w += LN) Wd.push_x[w] = 0.0f;  // ClearForces (at the end of Step; nothing reads it in between)
    MW_FLOPS(16 * NB);   // integrate velocities, integrate positions, sleep test
    MW_STAT(steps, 1); MW_STAT(sub_a, n_rounds); MW_STAT(sub_b, max_cnt); MW_STAT(manifolds, S.nm);
    MW_STAT(maxcnt_step, max_cnt); MW_STAT(cnt_hist[max_cnt < 23 ? max_cnt : 23], 1); MW_STAT(rounds_hist[n_rounds < 11 ? n_rounds : 11], 1);
    constexpr int NREG = Par::MREG > 0 ? Par::MREG : 1;
    SolveLane<NREG> LS[Par::SOLVE_EMU];
#define MW_LANES for (int li_ = 0; li_ < Par::SOLVE_EMU; ++li_)
#define MW_LANE const int sl_ = par.solve_lane(li_); SolveLane<NREG> &ls = LS[Par::SOLVE_EMU == 1 ? 0 : sl_]; (void)sl_;
    MW_LANES { MW_LANE
        ls.cnt = S.lane_cnt[sl_];
        MW_UNROLL
        for (int r = 0; r < NREG; ++r) {
            ls.mrd[r] = -1; ls.mix[r] = -1;
            if (r < Par::MREG && r < ls.cnt) {
                const int k = S.lane_list[sl_][r];
                ls.mix[r] = k; ls.mrd[r] = S.m_round[k]; ls.mc[r] = MP[k];
                ls.mq[r] = mass_of_pair(S, ls.mc[r].bA, ls.mc[r].bB);
            }
        }
        ls.jn = sl_ < NW ? S.jn[sl_] : 0;
        MW_UNROLL
        for (int q = 0; q < 4; ++q) { ls.ji[q] = q < ls.jn ? S.jorder[sl_][q] : 0; ls.jisl[q] = q < ls.jn ? S.j_island[ls.ji[q]] : 0; }
        // the lane's manifolds past its private copies: slots of MO in lane order, then (no room) in place
        ls.mo_base = 0;
        for (int l2 = 0; l2 < sl_; ++l2) { const int c2 = S.lane_cnt[l2]; if (c2 > Par::MREG) ls.mo_base += c2 - Par::MREG; }
        for (int r = Par::MREG; r < ls.cnt; ++r) { const int o = ls.mo_base + r - Par::MREG; if (o < mo_cap) MO[o] = MP[S.lane_list[sl_][r]]; }
    }
    Manifold *MP_ = MP;   // (re-pointed after the velocity iterations, see par.pool_again below)
    MW_LDS Manifold *const MOL = (MW_LDS Manifold *)MO;
#define MW_POOL_MANIFOLD(r_) (ls.mo_base + (r_) - Par::MREG < mo_cap ? MO[ls.mo_base + (r_) - Par::MREG] : MP_[S.lane_list[sl_][r_]])
    // One contact sweep: round by round, inside a round position by position of the lanes' lists, all lanes at once (build_islands made
    // sure two constraints of one (round, position) never share a body, and that the order of any two that do is the island's).  F_ is
    // a statement over the manifold `m_` and its mass data `q_`.
    // Past the lane-private copies a manifold sits in MO (LDS) or, when that is full, in the pool (HBM).  POOL_ says how F_ gets at it:
    //   MW_POOL_IN_PLACE   a reference to wherever it sits (InitializeVelocityConstraints: once per step, writes most of it);
    //   MW_POOL_VELOCITY   a register copy of the part the velocity sweeps read, fetched in one go at the top of the position; the four
    //                      accumulated impulses go back afterwards.  (In place, the 180 sweeps went through a pointer that may be LDS or
    //                      HBM -- flat loads one field at a time, re-fetched after every store through the same pointer: a position past
    //                      the register copies cost 2.5 times one inside them.)
    //   MW_POOL_POSITION   a register copy of the part the position sweeps read; nothing goes back.
#define MW_POOL_IN_PLACE(F_)  { Manifold &m_ = MW_POOL_MANIFOLD(r_); const MassAB q_ = mass_of_pair(S, m_.bA, m_.bB); F_; }
#define MW_POOL_VELOCITY(F_)  {                                                                                           \
        const int o_ = ls.mo_base + r_ - Par::MREG;                                                                       \
        Manifold m_;                                                                                                      \
        if (o_ < mo_cap) manifold_velocity_part_from(m_, MOL + o_); else manifold_velocity_part_from(m_, MP_ + k_);       \
        const MassAB q_ = mass_of_pair(S, m_.bA, m_.bB);                                                                  \
        F_;                                                                                                               \
        if (o_ < mo_cap) manifold_store_impulses_to(MOL + o_, m_); else manifold_store_impulses_to(MP_ + k_, m_);         \
    }
#define MW_POOL_POSITION(F_)  {                                                                                           \
        const int o_ = ls.mo_base + r_ - Par::MREG;                                                                       \
        Manifold m_;                                                                                                      \
        if (o_ < mo_cap) manifold_position_part_from(m_, MOL + o_); else manifold_position_part_from(m_, MP_ + k_);       \
        const MassAB q_ = mass_of_pair(S, m_.bA, m_.bB);                                                                  \
        F_;                                                                                                               \
    }
#define MW_CONTACT_SWEEP(F_, POOL_)                                                                                       \
    for (int rd_ = 0; rd_ < n_rounds; ++rd_) {                                                                            \
        MW_UNROLL                                                                                                         \
        for (int r_ = 0; r_ < NREG; ++r_) {                                                                               \
            if (r_ >= Par::MREG || r_ >= max_cnt) continue;                                                               \
            MW_LANES { MW_LANE if (ls.mrd[r_] == rd_) { Manifold &m_ = ls.mc[r_]; const MassAB &q_ = ls.mq[r_]; F_; } }    \
            par.sync();                                                                                                   \
        }                                                                                                                 \
        for (int r_ = Par::MREG; r_ < max_cnt; ++r_) {   /* past the lane-private copies */                               \
            MW_LANES { MW_LANE                                                                                            \
                if (r_ >= ls.cnt) continue;                                                                               \
                const int k_ = S.lane_list[sl_][r_];                                                                      \
                if (S.m_round[k_] != rd_) continue;                                                                       \
                POOL_(F_)                                                                                                 \
            }                                                                                                             \
            par.sync();                                                                                                   \
        }                                                                                                                 \
    }
    // ---- contact constraints: b2ContactSolver::InitializeVelocityConstraints + WarmStart, in the island's order
    MW_TSTAMP(0, 1); MW_TVAL(0, 0, max_cnt); MW_TVAL(0, 1, n_rounds);
    MW_CONTACT_SWEEP(contact_init_warm(Wd, m_, q_), MW_POOL_IN_PLACE)
    // ---- joints: InitVelocityConstraints (+ warm start)
    MW_LANES { MW_LANE
        MW_UNROLL
        for (int q = 0; q < 4; ++q) if (q < ls.jn) joint_init_warm(M, Wd, Cd, S, ls.ji[q], h, ls.jc[q]);
    }
    par.sync();
    MW_TSTAMP(0, 2);
    // ---- velocity iterations: all joints, then all contacts
    for (int it = 0; it < VEL_ITERS; ++it) {
        MW_LANES { MW_LANE
            MW_UNROLL
            for (int q = 0; q < 4; ++q) if (q < ls.jn) joint_solve_velocity(Wd, ls.jc[q]);
        }
        par.sync();
        MW_CONTACT_SWEEP(contact_solve_velocity(Wd, m_, q_), MW_POOL_VELOCITY)
    }
    MW_TSTAMP(0, 3);
    // From here on the record is addressed through pointers found again from the lane id (par.cold_again / pool_again: the identity
    // everywhere but in the one-launch kernel of the sixteen-lane class): no per-lane pointer is then live across the sweeps above, which
    // is where hipcc 7.2 parked them under a narrowed exec mask (multiwalker_impl.hpp, HAVE_FUSED).
    const ColdView CdW = par.cold_again(Cd);
    Manifold *const MPW = par.pool_again(MP);
    MP_ = MPW;
    // (the joint ids too: read again from the schedule instead of being carried over the sweeps -- one more value the join could park)
    MW_LANES { MW_LANE
        MW_UNROLL
        for (int q = 0; q < 4; ++q) ls.ji[q] = q < ls.jn ? S.jorder[sl_][q] : 0;
    }
    // the accumulated joint impulses and limit states go back to the world (warm start of the next step)
    MW_LANES { MW_LANE
        MW_UNROLL
        for (int q = 0; q < 4; ++q) {
            if (q >= ls.jn) continue;
            Joint &j = CdW.j[ls.ji[q]];
            const JointCache &c = ls.jc[q];
            j.ix = c.ix; j.iy = c.iy; j.iz = c.iz; j.motor_impulse = c.motor_impulse; j.limit_state = c.limit_state;
        }
    }
    // b2ContactSolver::StoreImpulses -> the contacts (warm start of the next step); every lane stores its own
    MW_LANES { MW_LANE
        MW_UNROLL
        for (int r = 0; r < NREG; ++r) {
            if (r >= Par::MREG || ls.mix[r] < 0) continue;
            const Manifold &m = ls.mc[r];
            Slot &sl = CdW.slot[m.slot];
            MW_UNROLL
            for (int i = 0; i < 2; ++i) if (i < m.npts) { sl.ni[i] = m.ni[i]; sl.ti[i] = m.ti[i]; }   // (constant indices: mc stays in registers)
        }
        for (int r = Par::MREG; r < ls.cnt; ++r) {
            const Manifold &m = MW_POOL_MANIFOLD(r);
            Slot &sl = CdW.slot[m.slot];
            for (int i = 0; i < m.npts; ++i) { sl.ni[i] = m.ni[i]; sl.ti[i] = m.ti[i]; }
        }
    }
    // The position iterations need the other half of every constraint (the b2Manifold in local coordinates, the joints' local anchors and
    // limits), the velocity iterations above did not: it is fetched again here instead of being carried through them in registers.
    MW_LANES { MW_LANE
        MW_UNROLL
        for (int r = 0; r < NREG; ++r) if (r < Par::MREG && ls.mix[r] >= 0) manifold_position_part(ls.mc[r], MPW[ls.mix[r]]);
        MW_UNROLL
        for (int q = 0; q < 4; ++q) if (q < ls.jn) joint_position_part(M, S, ls.ji[q], ls.jc[q]);
    }
    // ---- integrate positions
    for (int bi = L0; bi < NB; bi += LN) {
        if (S.island_of[bi] < 0) continue;
        Body &b = Wd.b[bi];
        V2 tr = h * b.v;
        if (dot(tr, tr) > MAX_TRANSLATION * MAX_TRANSLATION) { const float ratio = MAX_TRANSLATION / sqrtf(dot(tr, tr)); b.v = ratio * b.v; }
        const float ro = h * b.w;
        if (ro * ro > MAX_ROTATION * MAX_ROTATION) { const float ratio = MAX_ROTATION / fabsf(ro); b.w *= ratio; }
        b.c = b.c + h * b.v;
        b.a += h * b.w;
    }
    par.sync();
    // ---- position iterations: contacts then joints; each island stops on its own (b2Island::Solve early exit)
    // b2ContactSolver::SolvePositionConstraints returns min separation >= -3 linearSlop over the island's contacts -- true exactly when
    // every one of them is; b2Island::Solve ands that with every joint's answer.  So each lane collects one "not yet" bit per island
    // from the constraints it runs, the lanes of the env OR their words together (registers; no per-body / per-joint flags in LDS, no
    // serial pass over the islands), and every lane carries the same word of finished islands.
    const int n_isl = S.n_isl;
    const uint32_t all_isl = (1u << n_isl) - 1u;
    uint32_t isl_done = 0;
    MW_TSTAMP(0, 4);
    int pos_its_ = 0; (void)pos_its_;
    for (int it = 0; it < POS_ITERS; ++it) {
        pos_its_ = it + 1;
        MW_STAT(pos_iters, 1); MW_STAT(pos_iters_step, 1);
        uint32_t not_yet = 0;
        MW_CONTACT_SWEEP(if (!((isl_done >> m_.island) & 1u)) { if (!(contact_solve_position(Wd, m_, q_) >= -3.0f * LINEAR_SLOP)) not_yet |= 1u << m_.island; }, MW_POOL_POSITION)
        MW_LANES { MW_LANE
            MW_UNROLL
            for (int q = 0; q < 4; ++q) {
                if (q >= ls.jn) continue;
                const int jisl = ls.jisl[q];
                if ((isl_done >> jisl) & 1u) continue;
                if (!joint_solve_position(Wd, ls.jc[q])) not_yet |= 1u << jisl;
            }
        }
        par.sync();
        isl_done |= ~par.reduce_or(not_yet) & all_isl;
        if (isl_done == all_isl) break;
    }
    if (L0 == 0) for (int c = 0; c < n_isl; ++c) S.isl_pos_solved[c] = (uint8_t)((isl_done >> c) & 1u);   // positionSolved: stopped before the iterations ran out
    par.sync();
    MW_TSTAMP(0, 5); MW_TVAL(0, 2, pos_its_);
#undef MW_CONTACT_SWEEP
#undef MW_POOL_IN_PLACE
#undef MW_POOL_VELOCITY
#undef MW_POOL_POSITION
#undef MW_POOL_MANIFOLD
#undef MW_LANES
#undef MW_LANE
    // ---- sleeping (b2Island::Solve, allowSleep): an island whose slowest-to-rest body has been below the tolerances for half a
    // second, and whose position constraints were solved, goes to sleep: velocities zeroed, awake flags cleared
    if (L0 == 0) {
        for (int c = 0; c < n_isl; ++c) {
            float min_sleep = 3.402823466e+38f;
            for (int bi = 0; bi < NB; ++bi) {
                if (S.island_of[bi] != c) continue;
                const Body &b = Wd.b[bi];
                if (b.w * b.w > ANGULAR_SLEEP_TOLERANCE * ANGULAR_SLEEP_TOLERANCE || dot(b.v, b.v) > LINEAR_SLEEP_TOLERANCE * LINEAR_SLEEP_TOLERANCE) {
                    CdW.sleep_time[bi] = 0.0f; min_sleep = 0.0f;
                } else {
                    CdW.sleep_time[bi] += h;
                    min_sleep = mnf(min_sleep, CdW.sleep_time[bi]);
                }
            }
            if (min_sleep >= TIME_TO_SLEEP && S.isl_pos_solved[c])
                for (int bi = 0; bi < NB; ++bi) {
                    if (S.island_of[bi] != c) continue;
                    Wd.awake.clear(bi); CdW.sleep_time[bi] = 0.0f;
                    Wd.b[bi].v = v2(0, 0); Wd.b[bi].w = 0.0f;
                }
        }
        Wd.batch += 1;   // this step's FindNewContacts call (step_post)
        // A contact remembers its call in 16 bits (Slot::batch): past 65 535 calls in ONE episode -- some 45 000 steps of walking, 9 000 with
        // ten fallen walkers and half a dozen continuous-pass events per step (found by scripts/mw_soak.py --gait 0.9 --no-terminate at ten
        // walkers, step 9 141) -- new contacts would sort before old ones.  Said in-band instead (sticky bit 3, a margin of one step's events early).
        if (Wd.batch >= 0xFF00u) Wd.overflow |= 8;
    }
    par.sync();
}

// The tail of b2World::Solve: SynchronizeFixtures of the simulated bodies, then FindNewContacts for the proxies that moved
template <class Par>
MW_HD_INLINE void step_post(const Model &M, Hot &Wd, const ColdView &Cd, Scratch &S, Par par) {
    const int L0 = par.lane(), LN = par.n(), NB = M.NB;
    {
        BodyBits mv = BodyBits::none();
        for (int bi = L0; bi < NB; bi += LN)
            if (S.island_of[bi] >= 0 && sync_fixture(M, Wd, Cd, bi)) { mv.set(bi); find_new_terrain_contacts(M, Wd, Cd, bi, Wd.batch); }
        if (mv.any()) par.or_bits(&S.moved, mv);
    }
    par.sync();
    if (L0 == 0 && S.moved.any()) find_new_pair_contacts(M, Wd, Cd, S.moved, Wd.batch);
    par.sync();
}

template <class Par>
MW_HD_INLINE void world_step(const Model &M, Hot &Wd, const ColdView &Cd, Scratch &S, Par par) {
    MW_PHASE(0);
    step_collide(M, Wd, Cd, S, S.m, par);
    MW_PHASE(1);
    step_solve(M, Wd, Cd, S, S.m, par.solve_overflow(), Par::SOLVE_OVERFLOW, par);
    MW_PHASE(2);
    step_post(M, Wd, Cd, S, par);
    MW_PHASE(3);
    // ---- continuous pass (b2World::Step: "if (m_continuousPhysics && step.dt > 0) SolveTOI(step)")
    if (M.continuous) {
        ToiWork T;
        float toi_alpha[EDGE_SLOTS_PKG_MAX]; uint8_t toi_meta[EDGE_SLOTS_PKG_MAX];
        ToiLaneWork TL; TL.alpha = toi_alpha; TL.meta = toi_meta; TL.ovf = S.m; TL.ovf_cap = M.max_manifolds;
        solve_toi(M, Wd, Cd, S, T, TL, par, 1.0f / FPS);
    }
}


// ---------------------------------------------------------------- lidar: b2World::RayCast -> b2EdgeShape::RayCast over the terrain, closest hit (D2)
MW_HD float lidar_fraction(const Model &M, const ColdView &Cd, V2 p1, V2 p2) {
    MW_FLOPS(12);
    const V2 d = p2 - p1;
    float best = 1.0f;  // LidarCallback.fraction starts at 1.0 (:210)
    int e0 = (int)floorf(mnf(p1.x, p2.x) / TERRAIN_STEP) - 1, e1 = (int)floorf(mxf(p1.x, p2.x) / TERRAIN_STEP) + 1;
    if (e0 < 0) e0 = 0;
    if (e1 > M.NT - 2) e1 = M.NT - 2;
    for (int e = e0; e <= e1; ++e) {
        MW_FLOPS(30);   // b2EdgeShape::RayCast
        const V2 v1 = v2(M.tx[e], Cd.ty[e]), v2e = v2(M.tx[e + 1], Cd.ty[e + 1]);
        const V2 ee = v2e - v1;
        V2 normal = v2(ee.y, -ee.x);
        { const float inv = 1.0f / sqrtf(dot(normal, normal)); normal.x *= inv; normal.y *= inv; }
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
MW_HD double u24d(uint32_t r) { return (double)(r >> 8) * (1.0 / 16777216.0); }

template <class Par>
MW_HD void env_observe(const Model &M, const EnvCfg &C, Hot &Wd, const ColdView &Cd, Par par, uint32_t gid, float *obs, float *rew, uint8_t *done, double *rw);

// MultiWalkerEnv.reset (:330-357) without its trailing step: a fresh b2World (D1) with the package, the terrain edges and the
// walkers created in the reference's order.  terrain_in (NT float64 heights) / push_in (W float64) replace the Philox draws (D3).
MW_HD void env_reset_world(const Model &M, const EnvCfg &C, Hot &Wd, const ColdView &Cd, uint32_t gid, const double *terrain_in = nullptr, const double *push_in = nullptr) {
    // D3: the draws of a reset are keyed by (env, episode) -- the world an env gets is a function of how many episodes it has had, not of
    // when the previous one ended; that is what lets the HIP build prepare the next episode's world ahead of time
    const uint32_t tick = Wd.episode;
    Wd.game_over = 0; Wd.overflow = 0; Wd.prev_package_shaping = 0.0; Wd.t = 0;
    for (int w = 0; w < MAX_WALKERS; ++w) { Wd.fallen[w] = 0; Wd.prev_shaping[w] = 0.0; Wd.ground[w][0] = Wd.ground[w][1] = 0; }
    // every field, also of the empty entries: a record after reset is a function of (env, episode) alone, whatever was in its bytes before
    // (the HIP build prepares it in a spare record and copies it over the live one)
    for (int k = 0; k < M.n_slots; ++k) { Slot &sl = Cd.slot[k]; sl.edge = -1; sl.npts = 0; sl.touching = 0; sl.id[0] = sl.id[1] = 0; sl.ni[0] = sl.ni[1] = sl.ti[0] = sl.ti[1] = 0.0f; sl.batch = 0; sl.reserved_ = 0; }
    // _generate_terrain, non-hardcore branch (:516-612): float64 like the reference's Python loop, float32 when it enters Box2D
    {
        double velocity = 0.0, y = TERRAIN_HEIGHT64;
        int counter = TERRAIN_STARTPAD;
        bool oneshot = false;
        for (int i = 0; i < M.NT; ++i) {
            uint32_t r[4];
            philox10(gid, tick, (uint32_t)i, TAG_MW_TERRAIN, C.k0, C.k1, r);
            if (!oneshot) {
                const double d = TERRAIN_HEIGHT64 - y;
                velocity = 0.8 * velocity + 0.01 * (d > 0.0 ? 1.0 : (d < 0.0 ? -1.0 : 0.0));
                if (i > TERRAIN_STARTPAD) velocity += (2.0 * u24d(r[0]) - 1.0) / 30.0;  // np_random.uniform(-1, 1) / SCALE
                y += velocity;
            }
            oneshot = false;
            Cd.ty[i] = (float)(terrain_in ? terrain_in[i] : y);
            counter -= 1;
            if (counter == 0) {
                counter = TERRAIN_GRASS / 2 + (int)(((uint64_t)r[1] * (uint64_t)(TERRAIN_GRASS - TERRAIN_GRASS / 2)) >> 32);  // randint(5, 10)
                oneshot = true;
            }
        }
    }
    // _generate_package (:499-514)
    {
        Body &b = Wd.b[0];
        b.a = 0.0f; b.v = v2(0, 0); b.w = 0.0f;
        b.c = v2((float)M.mean_start_x64, (float)(TERRAIN_HEIGHT64 + 3 * LEG_H64)) + M.shape[SH_PACKAGE].centroid;
    }
    // BipedalWalker._reset (:113-192)
    const double init_y = TERRAIN_HEIGHT64 + 2 * LEG_H64;
    for (int w = 0; w < M.W; ++w) {
        const float init_x = M.start_x[w];
        Body &hull = Wd.b[hull_of(w)];
        hull.a = 0.0f; hull.v = v2(0, 0); hull.w = 0.0f;
        hull.c = v2(init_x, (float)init_y) + M.shape[SH_HULL].centroid;
        double push;
        if (push_in) push = push_in[w];
        else { uint32_t r[4]; philox10(gid, tick, (uint32_t)w, TAG_MW_PUSH, C.k0, C.k1, r); push = (2.0 * u24d(r[0]) - 1.0) * 5.0; }  // uniform(-INITIAL_RANDOM, INITIAL_RANDOM)
        Wd.push_x[w] = (float)push;
        for (int side = 0; side < 2; ++side) {
            const double sgn = side == 0 ? -1.0 : 1.0;
            Body &up = Wd.b[hull_of(w) + 1 + 2 * side], &lo = Wd.b[hull_of(w) + 2 + 2 * side];
            up.a = (float)(sgn * 0.05); up.v = v2(0, 0); up.w = 0.0f;
            up.c = v2(init_x, (float)(init_y - LEG_H64 / 2 - LEG_DOWN64)) + mul(rot(up.a), M.shape[SH_UPPER].centroid);
            lo.a = (float)(sgn * 0.05); lo.v = v2(0, 0); lo.w = 0.0f;
            lo.c = v2(init_x, (float)(init_y - LEG_H64 * 3 / 2 - LEG_DOWN64)) + mul(rot(lo.a), M.shape[SH_LOWER].centroid);
            Joint &hip = Cd.j[4 * w + 2 * side], &knee = Cd.j[4 * w + 2 * side + 1];
            hip.ix = hip.iy = hip.iz = hip.motor_impulse = 0.0f; hip.limit_state = 0;
            hip.motor_speed = (float)sgn; hip.max_torque = MOTORS_TORQUE;
            knee.ix = knee.iy = knee.iz = knee.motor_impulse = 0.0f; knee.limit_state = 0;
            knee.motor_speed = 1.0f; knee.max_torque = MOTORS_TORQUE;
        }
    }
    // broad phase: every proxy is created with its tight box fattened by b2_aabbExtension and buffered as moved; the first Step
    // begins with FindNewContacts (e_newFixture): batch 0
    Wd.awake = BodyBits::first(M.NB);
    Wd.batch = 0;
    for (int b = 0; b < M.NB; ++b) {
        Cd.sleep_time[b] = 0.0f;
        Cd.sweep_c0[b] = Wd.b[b].c; Cd.sweep_a0[b] = Wd.b[b].a; Cd.sweep_alpha0[b] = 0.0f;
        set_body_fat(Cd, b, fatten(poly_aabb(M.shape[shape_of_body(b)], body_xf(M, Wd.b[b], b))));
    }
    for (int b = 0; b < M.NB; ++b) find_new_terrain_contacts(M, Wd, Cd, b, 0);
    find_new_pair_contacts(M, Wd, Cd, BodyBits::first(M.NB), 0);
    Wd.episode = tick + 1;
    Wd.tick = 0;
}

// MultiWalkerEnv.step (:359-428).  obs: [W][32], rew: [W]
// apply_action (:194-203) for all walkers: the first thing MultiWalkerEnv.step does
template <class Par>
MW_HD_INLINE void env_apply_actions(const Model &M, Hot &Wd, const ColdView &Cd, Par par, const float *actions) {
    for (int w = par.lane(); w < M.W; w += par.n()) {
        for (int k = 0; k < 4; ++k) {
            const float a = actions[4 * w + k];
            Joint &j = Cd.j[4 * w + k];
            const float sp = (k % 2 == 0) ? SPEED_HIP : SPEED_KNEE;
            j.motor_speed = sp * (a > 0.0f ? 1.0f : (a < 0.0f ? -1.0f : 0.0f));
            j.max_torque = MOTORS_TORQUE * clampf(fabsf(a), 0.0f, 1.0f);
        }
    }
    if (par.lane() == 0) {   // b2RevoluteJoint::SetMotorSpeed / SetMaxMotorTorque wake both bodies of every joint: all the walkers' bodies
        for (int b = 1; b < M.NB; ++b) if (!Wd.awake.test(b)) { Wd.awake.set(b); Cd.sleep_time[b] = 0.0f; }
    }
    par.sync();
}
template <class Par>
MW_HD_INLINE void env_step(const Model &M, const EnvCfg &C, Hot &Wd, const ColdView &Cd, Scratch &S, Par par, uint32_t gid, const float *actions, float *obs,
                    float *rew, uint8_t *done) {
    env_apply_actions(M, Wd, Cd, par, actions);
    world_step(M, Wd, Cd, S, par);  // :365
    double rw[MAX_WALKERS];
    MW_PHASE(5);
    env_observe(M, C, Wd, Cd, par, gid, obs, rew, done, rw);
    if (par.lane() == 0) {
        Wd.t += 1;
        Wd.tick += 1;
    }
    par.sync();
}

// get_observation + the reward / done logic of MultiWalkerEnv.step: the walkers' rows by lane (walker w on lane w % n), then one lane
// for what couples them.  rw: MAX_WALKERS doubles the lanes share (LDS in the HIP kernel).  No indexed local arrays (they would live
// in scratch memory on the GPU): hull positions and noise pairs are recomputed where they are needed.
template <class Par>
MW_HD void env_observe(const Model &M, const EnvCfg &C, Hot &Wd, const ColdView &Cd, Par par, uint32_t gid, float *obs, float *rew, uint8_t *done, double *rw) {
    // the Python side of the reference computes in float64 on the float32 values Box2D hands it; so does this function
    const Body &pkg = Wd.b[0];
    const V2 pkg_pos = body_xf(M, pkg, 0).p;
    const double package_length = 240.0 / 30.0 * (M.W / 1.75);   // :293-294
    const bool noisy = C.position_noise != 0.0f || C.angle_noise != 0.0f;
    for (int w = par.lane(); w < M.W; w += par.n()) {
        MW_FLOPS(90 + 40 / M.W);   // besides the lidar: transforms of the hulls and the package, the 14 + 8 state entries, shaping, rewards
        const Body &hull = Wd.b[hull_of(w)];
        const V2 pos = body_xf(M, hull, hull_of(w)).p;
        float *o = obs + w * obs_dim_of(C);
        // get_observation (:205-237)
        o[0] = hull.a;
        o[1] = (float)(2.0 * (double)hull.w / 50.0);
        o[2] = (float)(0.3 * (double)hull.v.x * (600.0 / 30.0) / 50.0);
        o[3] = (float)(0.3 * (double)hull.v.y * (400.0 / 30.0) / 50.0);
        for (int side = 0; side < 2; ++side) {
            const Body &up = Wd.b[hull_of(w) + 1 + 2 * side], &lo = Wd.b[hull_of(w) + 2 + 2 * side];
            o[4 + 5 * side + 0] = up.a - hull.a;                                        // joints[0/2].angle (float32 in Box2D)
            o[4 + 5 * side + 1] = (float)((double)(up.w - hull.w) / 4.0);               // .speed / SPEED_HIP
            o[4 + 5 * side + 2] = (float)((double)(lo.a - up.a) + 1.0);                 // joints[1/3].angle + 1.0
            o[4 + 5 * side + 3] = (float)((double)(lo.w - up.w) / 6.0);
            o[4 + 5 * side + 4] = Wd.ground[w][side] ? 1.0f : 0.0f;
        }
        for (int i = 0; i < 10; ++i) {  // lidar (:209-214): p2 in float64, float32 when it enters RayCast
            const V2 p2 = v2((float)((double)pos.x + M.lidar_dx[i]), (float)((double)pos.y - M.lidar_dy[i]));
            o[14 + i] = lidar_fraction(M, Cd, pos, p2);
        }
        // neighbours and package (:380-400), gaussian noise via Box-Muller on keyed uniforms: draw q gives the values 2q and 2q + 1 of the
        // walker's seven normals (:389-395 consumes them in order: left neighbour, right neighbour, package x, y, angle)
        auto normal_pair = [&](int q, float &a, float &b) {
            a = 0.0f; b = 0.0f;
            if (!noisy) return;
            uint32_t r[4];
            philox10(gid, Wd.episode, (Wd.tick << 6) | (uint32_t)(w * 4 + q), TAG_MW_NOISE, C.k0, C.k1, r);
            const float u1 = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f), u2 = u24f(r[1]);
            const float rad = sqrtf(-2.0f * logf(u1));
            float bs, bc;
            sincos_det(2.0f * B2_PI * u2, bs, bc);
            a = rad * bc; b = rad * bs;
        };
        int n = 24, zi = 0;
        for (int dj = -1; dj <= 1; dj += 2) {
            const int j = w + dj;
            if (j < 0 || j == M.W) { o[n++] = 0.0f; o[n++] = 0.0f; }
            else {
                float za, zb;
                normal_pair(zi, za, zb); ++zi;
                const V2 hj = body_xf(M, Wd.b[hull_of(j)], hull_of(j)).p;
                const double xm = ((double)hj.x - (double)pos.x) / package_length, ym = ((double)hj.y - (double)pos.y) / package_length;
                o[n++] = (float)(xm + (double)C.position_noise * (double)za);
                o[n++] = (float)(ym + (double)C.position_noise * (double)zb);
            }
        }
        float z4, z5, z6, z7;
        normal_pair(2, z4, z5); normal_pair(3, z6, z7);
        const double xd = ((double)pkg_pos.x - (double)pos.x) / package_length, yd = ((double)pkg_pos.y - (double)pos.y) / package_length;
        o[n++] = (float)(xd + (double)C.position_noise * (double)z4);
        o[n++] = (float)(yd + (double)C.position_noise * (double)z5);
        o[n++] = (float)((double)pkg.a + (double)C.angle_noise * (double)z6);
        if (C.one_hot) { for (int k = 0; k < MAX_AGENTS_ID; ++k) o[n++] = (k == w) ? 1.0f : 0.0f; }  // np.eye(MAX_AGENTS)[i] :397-398
        else o[n++] = (float)((double)w / (double)M.W);  // :400
        // shaping (:403-407)
        const double shaping = 0.0 - 5.0 * fabs((double)hull.a);
        rw[w] = shaping - Wd.prev_shaping[w];
        Wd.prev_shaping[w] = shaping;
    }
    par.sync();
    if (par.lane() != 0) return;
    const double package_shaping = (double)C.forward_reward * 130 * (double)pkg_pos.x / 30.0;  // :409-411
    for (int w = 0; w < M.W; ++w) rw[w] += (package_shaping - Wd.prev_package_shaping);
    Wd.prev_package_shaping = package_shaping;
    bool dn = false;
    const double last_x = (double)body_xf(M, Wd.b[hull_of(M.W - 1)], hull_of(M.W - 1)).p.x;  // `pos` leaks out of the loop: the LAST walker (:417, :420)
    if (Wd.game_over || last_x < 0.0) { for (int w = 0; w < M.W; ++w) rw[w] += (double)C.drop_reward; dn = true; }
    if (last_x > (M.NT - TERRAIN_GRASS) * (14.0 / 30.0)) dn = true;
    int nfallen = 0;
    for (int w = 0; w < M.W; ++w) { rw[w] += (double)C.fall_reward * (Wd.fallen[w] ? 1.0 : 0.0); nfallen += Wd.fallen[w]; }
    if (C.terminate_on_fall && nfallen > 0) dn = true;
    if (rew) {
        if (C.reward_global) { double s = 0.0; for (int w = 0; w < M.W; ++w) s += rw[w]; s /= (double)M.W; for (int w = 0; w < M.W; ++w) rew[w] = (float)s; }
        else for (int w = 0; w < M.W; ++w) rew[w] = (float)rw[w];
    }
    if (done) *done = dn ? 1 : 0;
}

}  // namespace MW_NS
