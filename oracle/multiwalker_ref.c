/*
 * multiwalker_ref.c -- INDEPENDENT CPU restatement of the reference MultiWalkerEnv
 * (/root/reference/madrl_environments/walker/multi_walker.py) together with the part of Box2D it drives
 * (`self.world.Step(1.0 / FPS, 6 * 30, 2 * 30)`, multi_walker.py:365).
 *
 * TEST INFRASTRUCTURE ONLY: the parity checker of the HIP kernels (tests/test_multiwalker_*.py).  Nothing under
 * madrl_amd/ includes, links or calls it, and -- unlike oracle/multiwalker_oracle.cpp, which compiles the PRODUCT's own
 * source with g++ and therefore only checks the GPU port -- this file shares no code with madrl_amd/csrc: plain scalar C,
 * float32, Box2D's own data structures (body / contact / joint linked lists, contact edges, b2Island arrays, b2Sweep).
 *
 * PARITY UNPINNED.  Box2D (pybox2d / box2d-py, Box2D 2.3.0 inside; no version pin anywhere in the reference tree) cannot be
 * imported, built or installed in this image and the reference holds no golden vectors at that boundary.  Everything that is
 * not a multi_walker.py citation restates Box2D 2.3.0's published algorithms FROM MEMORY of its source layout and must be
 * re-verified when a Box2D tree is at hand (oracle/make_golden_multiwalker.py records the unmodified reference the moment
 * `import Box2D` works).  The one published anchor, the six printed lines of the "Hello Box2D" manual page, is replayed by
 * mwr_helloworld() below (tests/test_multiwalker_cpu.py).
 *
 * What IS pinned (round 4): the multi_walker.py half of this file.  The `mwb_*` entry points at the end expose the bare World;
 * oracle/shims_box2d wraps them as a package named `Box2D`, over which the UNMODIFIED reference module imports and runs.  Its
 * recordings (oracle/make_golden_multiwalker.py --envlayer) are reproduced by mw_reset_world / mw_step free-running: the world the
 * reset constructs and every body state after it bit for bit, observations and rewards to 1e-12, ContactDetector flags and done
 * exactly, also with the observation noise and the reset draws scripted from the Philox contract (tests/test_multiwalker_envlayer.py).
 *
 * What follows Box2D 2.3.0 here (file names as in that tree):
 *   Dynamics/b2World.cpp         Step -> (FindNewContacts) -> Collide -> Solve (islands by DFS from the body list) -> SolveTOI
 *   Dynamics/b2ContactManager    AddPair / Destroy / Collide / FindNewContacts (fat AABBs, pairs sorted by proxy id)
 *   Dynamics/b2Island.cpp        Solve (sleeping included), SolveTOI
 *   Dynamics/Contacts/b2ContactSolver.cpp, b2Contact.cpp (Update: feature-id matching, Begin / EndContact)
 *   Dynamics/Joints/b2RevoluteJoint.cpp
 *   Collision/b2CollidePolygon.cpp (2.3.0: hill-climbing b2FindMaxSeparation, 0.98 / 0.001 hysteresis),
 *   b2CollideEdge.cpp (b2EPCollider), b2Collision.cpp (b2ClipSegmentToLine, b2WorldManifold), b2Distance.cpp (GJK),
 *   b2TimeOfImpact.cpp, Shapes/b2PolygonShape.cpp (Set, SetAsBox, ComputeMass, ComputeAABB), b2EdgeShape.cpp (RayCast)
 * Order is semantics and is Box2D's: new contacts are PREPENDED to the world list and to both bodies' edge lists; a
 * FindNewContacts batch creates its pairs sorted by (proxyIdA, proxyIdB); islands are built by depth-first search seeded from
 * the body list (last created body first) over contact edges, then joint edges; constraints are solved in island order.
 *
 * Stated differences from the real library (each also in DESIGN.md 4c):
 *   D1  b2DynamicTree is replaced by a linear scan over the proxies' FAT AABBs.  The pair SET, and after UpdatePairs' sort the
 *       pair ORDER, do not depend on the tree's shape; proxy ids are taken in creation order, as in a fresh b2World.  (The
 *       reference re-uses one b2World across reset()s, where ids come back from the tree's free list in an order that depends
 *       on the tree's shape at destroy time; every episode here starts like the first episode of a new MultiWalkerEnv.)
 *   D2  RayCast reports the CLOSEST terrain hit.  The reference's LidarCallback (:183-190) returns 0 on the first category-1
 *       fixture the tree traversal reports, i.e. the first hit in tree order, which equals the closest one whenever the ray
 *       crosses the terrain polyline once (b2EdgeShape::RayCast is two-sided; a second crossing needs a terrain slope above the
 *       ray's, > 3 sigma of the generator for the shallowest ray).
 *   D3  Randomness (terrain, initial push, observation noise) is this repo's counter-based Philox contract, not
 *       numpy's Mersenne Twister streams; parity runs take terrain and push from the caller.
 *
 * Reference map (multi_walker.py): constants :17-47; ContactDetector :50-84; BipedalWalker._reset :113-192; apply_action
 * :194-203; get_observation :205-237; setup / reset :276-357; step :359-428; _generate_package :499-514; _generate_terrain
 * :516-628 (hardcore = False, :254).
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fno-fast-math): every float operation rounds once, like Box2D built
 * without -ffast-math on SSE2.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------ b2Settings.h */
#define b2_pi 3.14159265359f
#define b2_epsilon FLT_EPSILON
#define b2_maxFloat FLT_MAX
#define b2_maxManifoldPoints 2
#define b2_maxPolygonVertices 8
#define b2_aabbExtension 0.1f
#define b2_aabbMultiplier 2.0f
#define b2_linearSlop 0.005f
#define b2_angularSlop (2.0f / 180.0f * b2_pi)
#define b2_polygonRadius (2.0f * b2_linearSlop)
#define b2_maxSubSteps 8
#define b2_maxTOIContacts 32
#define b2_velocityThreshold 1.0f
#define b2_maxLinearCorrection 0.2f
#define b2_maxAngularCorrection (8.0f / 180.0f * b2_pi)
#define b2_maxTranslation 2.0f
#define b2_maxTranslationSquared (b2_maxTranslation * b2_maxTranslation)
#define b2_maxRotation (0.5f * b2_pi)
#define b2_maxRotationSquared (b2_maxRotation * b2_maxRotation)
#define b2_baumgarte 0.2f
#define b2_toiBaugarte 0.75f
#define b2_timeToSleep 0.5f
#define b2_linearSleepTolerance 0.01f
#define b2_angularSleepTolerance (2.0f / 180.0f * b2_pi)

/* ------------------------------------------------------------------------------------------------ b2Math.h */
typedef struct { float x, y; } Vec2;
typedef struct { float s, c; } Rot;
typedef struct { Vec2 p; Rot q; } Xform;
typedef struct { Vec2 localCenter, c0, c; float a0, a, alpha0; } Sweep;

static inline Vec2 V(float x, float y) { Vec2 r; r.x = x; r.y = y; return r; }
static inline Vec2 vadd(Vec2 a, Vec2 b) { return V(a.x + b.x, a.y + b.y); }
static inline Vec2 vsub(Vec2 a, Vec2 b) { return V(a.x - b.x, a.y - b.y); }
static inline Vec2 vneg(Vec2 a) { return V(-a.x, -a.y); }
static inline Vec2 vscale(float s, Vec2 a) { return V(s * a.x, s * a.y); }
static inline float vdot(Vec2 a, Vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float vcross(Vec2 a, Vec2 b) { return a.x * b.y - a.y * b.x; }
static inline Vec2 vcross_vs(Vec2 a, float s) { return V(s * a.y, -s * a.x); }   /* b2Cross(v, s) */
static inline Vec2 vcross_sv(float s, Vec2 a) { return V(-s * a.y, s * a.x); }   /* b2Cross(s, v) */
static inline float vlen2(Vec2 a) { return a.x * a.x + a.y * a.y; }
static inline float vlen(Vec2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
static inline float vnormalize(Vec2 *a) {   /* b2Vec2::Normalize */
    const float length = vlen(*a);
    if (length < b2_epsilon) return 0.0f;
    const float inv = 1.0f / length;
    a->x *= inv; a->y *= inv;
    return length;
}
/* b2Min / b2Max / b2Clamp are plain comparisons in Box2D (b2Math.h): a < b ? a : b, a > b ? a : b, b2Max(low, b2Min(a, high)) */
static inline float b2minf(float a, float b) { return a < b ? a : b; }
static inline float b2maxf(float a, float b) { return a > b ? a : b; }
static inline float fclampf(float a, float lo, float hi) { return b2maxf(lo, b2minf(a, hi)); }

/* b2Rot::Set.  Box2D calls sinf / cosf; the kernels of this repo use a +,-,* polynomial so that their host and device builds
 * agree to the bit.  MWR_POLY_SINCOS restates that polynomial (Cody-Waite reduction by pi/2, single-precision minimax
 * polynomials on [-pi/4, pi/4]); without it this file uses libm like Box2D.  Both are within 1-2 ulp of the true
 * value. */
static inline Rot rot_of(float angle) {
    Rot q;
#ifdef MWR_POLY_SINCOS
    const float kf = floorf(angle * 0.636619772f + 0.5f);
    const int k = (int)kf;
    float r = (angle - kf * 1.5703125f) - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.549789948768648e-8f;
    const float z = r * r;
    const float ps = r + r * z * ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f);
    const float pc = (1.0f - 0.5f * z) + z * z * ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f);
    switch (k & 3) {
        case 0: q.s = ps; q.c = pc; break;
        case 1: q.s = pc; q.c = -ps; break;
        case 2: q.s = -ps; q.c = -pc; break;
        default: q.s = -pc; q.c = ps; break;
    }
#else
    q.s = sinf(angle); q.c = cosf(angle);
#endif
    return q;
}
static inline Vec2 rmul(Rot q, Vec2 v) { return V(q.c * v.x - q.s * v.y, q.s * v.x + q.c * v.y); }
static inline Vec2 rmulT(Rot q, Vec2 v) { return V(q.c * v.x + q.s * v.y, -q.s * v.x + q.c * v.y); }
static inline Vec2 xmul(Xform T, Vec2 v) { return V((T.q.c * v.x - T.q.s * v.y) + T.p.x, (T.q.s * v.x + T.q.c * v.y) + T.p.y); }
static inline Vec2 xmulT(Xform T, Vec2 v) {
    const float px = v.x - T.p.x, py = v.y - T.p.y;
    return V(T.q.c * px + T.q.s * py, -T.q.s * px + T.q.c * py);
}
static inline Rot rmulT_rr(Rot q, Rot r) { Rot o; o.s = q.c * r.s - q.s * r.c; o.c = q.c * r.c + q.s * r.s; return o; }
static inline Xform xmulT_xx(Xform A, Xform B) { Xform C; C.q = rmulT_rr(A.q, B.q); C.p = rmulT(A.q, vsub(B.p, A.p)); return C; }
static inline Xform xf_identity(void) { Xform T; T.p = V(0, 0); T.q.s = 0.0f; T.q.c = 1.0f; return T; }

static Xform sweep_transform(const Sweep *s, float beta) {   /* b2Sweep::GetTransform */
    Xform xf;
    xf.p = vadd(vscale(1.0f - beta, s->c0), vscale(beta, s->c));
    const float angle = (1.0f - beta) * s->a0 + beta * s->a;
    xf.q = rot_of(angle);
    xf.p = vsub(xf.p, rmul(xf.q, s->localCenter));
    return xf;
}
static void sweep_advance(Sweep *s, float alpha) {           /* b2Sweep::Advance */
    const float beta = (alpha - s->alpha0) / (1.0f - s->alpha0);
    s->c0 = vadd(s->c0, vscale(beta, vsub(s->c, s->c0)));
    s->a0 += beta * (s->a - s->a0);
    s->alpha0 = alpha;
}
static void sweep_normalize(Sweep *s) {                      /* b2Sweep::Normalize */
    const float twoPi = 2.0f * b2_pi;
    const float d = twoPi * floorf(s->a0 / twoPi);
    s->a0 -= d; s->a -= d;
}

/* ------------------------------------------------------------------------------------------------ shapes */
enum { SHAPE_POLYGON = 0, SHAPE_EDGE = 1 };
typedef struct {
    int type, count;                 /* polygon: count vertices; edge: vertices[0], vertices[1] (no ghost vertices: the
                                        reference builds edgeShape(vertices=[p1, p2]), :617-620, i.e. b2EdgeShape::Set(v1, v2)) */
    Vec2 v[b2_maxPolygonVertices], n[b2_maxPolygonVertices];
    Vec2 centroid;
    float radius;
} Shape;
typedef struct { Vec2 lo, hi; } AABB;

static void polygon_set_as_box(Shape *s, float hx, float hy) {   /* b2PolygonShape::SetAsBox(hx, hy) */
    s->type = SHAPE_POLYGON; s->count = 4; s->radius = b2_polygonRadius;
    s->v[0] = V(-hx, -hy); s->v[1] = V(hx, -hy); s->v[2] = V(hx, hy); s->v[3] = V(-hx, hy);
    s->n[0] = V(0.0f, -1.0f); s->n[1] = V(1.0f, 0.0f); s->n[2] = V(0.0f, 1.0f); s->n[3] = V(-1.0f, 0.0f);
    s->centroid = V(0, 0);
}
static Vec2 polygon_centroid(const Vec2 *vs, int count) {        /* ComputeCentroid (b2PolygonShape.cpp), pRef = origin */
    Vec2 c = V(0, 0);
    float area = 0.0f;
    const Vec2 pRef = V(0, 0);
    const float inv3 = 1.0f / 3.0f;
    for (int i = 0; i < count; ++i) {
        const Vec2 p1 = pRef, p2 = vs[i], p3 = i + 1 < count ? vs[i + 1] : vs[0];
        const Vec2 e1 = vsub(p2, p1), e2 = vsub(p3, p1);
        const float D = vcross(e1, e2), triangleArea = 0.5f * D;
        area += triangleArea;
        c = vadd(c, vscale(triangleArea * inv3, vadd(vadd(p1, p2), p3)));
    }
    return vscale(1.0f / area, c);
}
static void polygon_set(Shape *s, const Vec2 *vertices, int count) {   /* b2PolygonShape::Set: weld, gift-wrap hull, normals */
    s->type = SHAPE_POLYGON; s->radius = b2_polygonRadius;
    int n = count < b2_maxPolygonVertices ? count : b2_maxPolygonVertices;
    Vec2 ps[b2_maxPolygonVertices];
    int tempCount = 0;
    for (int i = 0; i < n; ++i) {
        const Vec2 v = vertices[i];
        int unique = 1;
        for (int j = 0; j < tempCount; ++j)
            if (vlen2(vsub(v, ps[j])) < 0.5f * b2_linearSlop) { unique = 0; break; }
        if (unique) ps[tempCount++] = v;
    }
    n = tempCount;
    int i0 = 0;
    float x0 = ps[0].x;
    for (int i = 1; i < n; ++i) {
        const float x = ps[i].x;
        if (x > x0 || (x == x0 && ps[i].y < ps[i0].y)) { i0 = i; x0 = x; }
    }
    int hull[b2_maxPolygonVertices], m = 0, ih = i0;
    for (;;) {
        hull[m] = ih;
        int ie = 0;
        for (int j = 1; j < n; ++j) {
            if (ie == ih) { ie = j; continue; }
            const Vec2 r = vsub(ps[ie], ps[hull[m]]), v = vsub(ps[j], ps[hull[m]]);
            const float c = vcross(r, v);
            if (c < 0.0f) ie = j;
            if (c == 0.0f && vlen2(v) > vlen2(r)) ie = j;   /* collinearity check */
        }
        ++m;
        ih = ie;
        if (ie == i0) break;
    }
    s->count = m;
    for (int i = 0; i < m; ++i) s->v[i] = ps[hull[i]];
    for (int i = 0; i < m; ++i) {
        const int i2 = i + 1 < m ? i + 1 : 0;
        const Vec2 edge = vsub(s->v[i2], s->v[i]);
        s->n[i] = vcross_vs(edge, 1.0f);
        vnormalize(&s->n[i]);
    }
    s->centroid = polygon_centroid(s->v, m);
}
typedef struct { float mass, I; Vec2 center; } MassData;
static MassData polygon_mass(const Shape *p, float density) {           /* b2PolygonShape::ComputeMass */
    Vec2 center = V(0, 0), s = V(0, 0);
    float area = 0.0f, I = 0.0f;
    for (int i = 0; i < p->count; ++i) s = vadd(s, p->v[i]);
    s = vscale(1.0f / p->count, s);
    const float k_inv3 = 1.0f / 3.0f;
    for (int i = 0; i < p->count; ++i) {
        const Vec2 e1 = vsub(p->v[i], s), e2 = vsub(i + 1 < p->count ? p->v[i + 1] : p->v[0], s);
        const float D = vcross(e1, e2), triangleArea = 0.5f * D;
        area += triangleArea;
        center = vadd(center, vscale(triangleArea * k_inv3, vadd(e1, e2)));
        const float ex1 = e1.x, ey1 = e1.y, ex2 = e2.x, ey2 = e2.y;
        const float intx2 = ex1 * ex1 + ex2 * ex1 + ex2 * ex2, inty2 = ey1 * ey1 + ey2 * ey1 + ey2 * ey2;
        I += (0.25f * k_inv3 * D) * (intx2 + inty2);
    }
    MassData md;
    md.mass = density * area;
    center = vscale(1.0f / area, center);
    md.center = vadd(center, s);
    md.I = density * I;
    md.I += md.mass * (vdot(md.center, md.center) - vdot(center, center));   /* shift to the body origin */
    return md;
}
static AABB shape_aabb(const Shape *s, Xform xf) {                       /* b2PolygonShape / b2EdgeShape::ComputeAABB */
    AABB b;
    const int n = s->type == SHAPE_EDGE ? 2 : s->count;
    Vec2 lower = xmul(xf, s->v[0]), upper = lower;
    for (int i = 1; i < n; ++i) {
        const Vec2 v = xmul(xf, s->v[i]);
        lower = V(b2minf(lower.x, v.x), b2minf(lower.y, v.y));
        upper = V(b2maxf(upper.x, v.x), b2maxf(upper.y, v.y));
    }
    b.lo = V(lower.x - s->radius, lower.y - s->radius);
    b.hi = V(upper.x + s->radius, upper.y + s->radius);
    return b;
}
static inline int aabb_overlap(const AABB *a, const AABB *b) {           /* b2TestOverlap(const b2AABB&, const b2AABB&) */
    const Vec2 d1 = vsub(b->lo, a->hi), d2 = vsub(a->lo, b->hi);
    if (d1.x > 0.0f || d1.y > 0.0f) return 0;
    if (d2.x > 0.0f || d2.y > 0.0f) return 0;
    return 1;
}
static inline int aabb_contains(const AABB *a, const AABB *b) {          /* b2AABB::Contains */
    return a->lo.x <= b->lo.x && a->lo.y <= b->lo.y && b->hi.x <= a->hi.x && b->hi.y <= a->hi.y;
}

/* ------------------------------------------------------------------------------------------------ manifolds */
enum { MF_FACE_A = 1, MF_FACE_B = 2 };          /* b2Manifold::e_faceA / e_faceB (e_circles unused) */
enum { CF_VERTEX = 0, CF_FACE = 1 };
typedef struct { uint8_t indexA, indexB, typeA, typeB; } Feature;
typedef struct { Vec2 localPoint; float normalImpulse, tangentImpulse; Feature id; } ManifoldPoint;
typedef struct { ManifoldPoint points[b2_maxManifoldPoints]; Vec2 localNormal, localPoint; int type, pointCount; } Manifold;
typedef struct { Vec2 v; Feature id; } ClipVertex;
static inline uint32_t feature_key(Feature f) { return (uint32_t)f.indexA | ((uint32_t)f.indexB << 8) | ((uint32_t)f.typeA << 16) | ((uint32_t)f.typeB << 24); }

static int clip_segment_to_line(ClipVertex vOut[2], const ClipVertex vIn[2], Vec2 normal, float offset, int vertexIndexA) {
    int numOut = 0;
    const float distance0 = vdot(normal, vIn[0].v) - offset, distance1 = vdot(normal, vIn[1].v) - offset;
    if (distance0 <= 0.0f) vOut[numOut++] = vIn[0];
    if (distance1 <= 0.0f) vOut[numOut++] = vIn[1];
    if (distance0 * distance1 < 0.0f) {
        const float interp = distance0 / (distance0 - distance1);
        vOut[numOut].v = vadd(vIn[0].v, vscale(interp, vsub(vIn[1].v, vIn[0].v)));
        vOut[numOut].id.indexA = (uint8_t)vertexIndexA;   /* VertexA is hitting edgeB */
        vOut[numOut].id.indexB = vIn[0].id.indexB;
        vOut[numOut].id.typeA = CF_VERTEX;
        vOut[numOut].id.typeB = CF_FACE;
        ++numOut;
    }
    return numOut;
}

/* b2CollidePolygon.cpp (2.3.0) */
static float edge_separation(const Shape *poly1, Xform xf1, int edge1, const Shape *poly2, Xform xf2) {
    const Vec2 normal1World = rmul(xf1.q, poly1->n[edge1]);
    const Vec2 normal1 = rmulT(xf2.q, normal1World);
    int index = 0;
    float minDot = b2_maxFloat;
    for (int i = 0; i < poly2->count; ++i) {
        const float d = vdot(poly2->v[i], normal1);
        if (d < minDot) { minDot = d; index = i; }
    }
    const Vec2 v1 = xmul(xf1, poly1->v[edge1]), v2 = xmul(xf2, poly2->v[index]);
    return vdot(vsub(v2, v1), normal1World);
}
static float find_max_separation(int *edgeIndex, const Shape *poly1, Xform xf1, const Shape *poly2, Xform xf2) {
    const int count1 = poly1->count;
    const Vec2 d = vsub(xmul(xf2, poly2->centroid), xmul(xf1, poly1->centroid));
    const Vec2 dLocal1 = rmulT(xf1.q, d);
    int edge = 0;
    float maxDot = -b2_maxFloat;
    for (int i = 0; i < count1; ++i) {
        const float dt = vdot(poly1->n[i], dLocal1);
        if (dt > maxDot) { maxDot = dt; edge = i; }
    }
    float s = edge_separation(poly1, xf1, edge, poly2, xf2);
    const int prevEdge = edge - 1 >= 0 ? edge - 1 : count1 - 1;
    const float sPrev = edge_separation(poly1, xf1, prevEdge, poly2, xf2);
    const int nextEdge = edge + 1 < count1 ? edge + 1 : 0;
    const float sNext = edge_separation(poly1, xf1, nextEdge, poly2, xf2);
    int bestEdge, increment;
    float bestSeparation;
    if (sPrev > s && sPrev > sNext) { increment = -1; bestEdge = prevEdge; bestSeparation = sPrev; }
    else if (sNext > s) { increment = 1; bestEdge = nextEdge; bestSeparation = sNext; }
    else { *edgeIndex = edge; return s; }
    for (;;) {   /* local search for the best edge normal */
        if (increment == -1) edge = bestEdge - 1 >= 0 ? bestEdge - 1 : count1 - 1;
        else edge = bestEdge + 1 < count1 ? bestEdge + 1 : 0;
        s = edge_separation(poly1, xf1, edge, poly2, xf2);
        if (s > bestSeparation) { bestEdge = edge; bestSeparation = s; }
        else break;
    }
    *edgeIndex = bestEdge;
    return bestSeparation;
}
/* b2CollidePolygon.cpp as it reads in LATER 2.3.x revisions: b2FindMaxSeparation visits every edge normal of poly1 (in poly2's frame) and keeps the
 * one whose deepest point of poly2 is the least deep -- no hill climbing, no centroids -- and b2CollidePolygons prefers poly1 = B only beyond
 * 0.1 * b2_linearSlop.  Which revision the authors' pybox2d wrapped is not recorded anywhere: World::polygonRevision
 * selects (0 = 2.3.0 above, the default; 1 = this one), so that whoever pins the dynamics against a real Box2D can try both. */
static float find_max_separation_all_edges(int *edgeIndex, const Shape *poly1, Xform xf1, const Shape *poly2, Xform xf2) {
    const Xform xf = xmulT_xx(xf2, xf1);
    int bestIndex = 0;
    float maxSeparation = -b2_maxFloat;
    for (int i = 0; i < poly1->count; ++i) {
        const Vec2 n = rmul(xf.q, poly1->n[i]);            /* poly1 normal in frame 2 */
        const Vec2 v1 = xmul(xf, poly1->v[i]);
        float si = b2_maxFloat;                            /* deepest point for normal i */
        for (int j = 0; j < poly2->count; ++j) {
            const float sij = vdot(n, vsub(poly2->v[j], v1));
            if (sij < si) si = sij;
        }
        if (si > maxSeparation) { maxSeparation = si; bestIndex = i; }
    }
    *edgeIndex = bestIndex;
    return maxSeparation;
}
static void collide_polygons(Manifold *m, const Shape *polyA, Xform xfA, const Shape *polyB, Xform xfB, int revision) {
    m->pointCount = 0;
    const float totalRadius = polyA->radius + polyB->radius;
    int edgeA = 0, edgeB = 0;
    const float separationA = revision ? find_max_separation_all_edges(&edgeA, polyA, xfA, polyB, xfB) : find_max_separation(&edgeA, polyA, xfA, polyB, xfB);
    if (separationA > totalRadius) return;
    const float separationB = revision ? find_max_separation_all_edges(&edgeB, polyB, xfB, polyA, xfA) : find_max_separation(&edgeB, polyB, xfB, polyA, xfA);
    if (separationB > totalRadius) return;
    const Shape *poly1, *poly2;
    Xform xf1, xf2;
    int edge1, flip;
    const float k_relativeTol = 0.98f, k_absoluteTol = 0.001f;
    if (revision ? (separationB > separationA + 0.1f * b2_linearSlop) : (separationB > k_relativeTol * separationA + k_absoluteTol)) {
        poly1 = polyB; poly2 = polyA; xf1 = xfB; xf2 = xfA; edge1 = edgeB; m->type = MF_FACE_B; flip = 1;
    } else {
        poly1 = polyA; poly2 = polyB; xf1 = xfA; xf2 = xfB; edge1 = edgeA; m->type = MF_FACE_A; flip = 0;
    }
    ClipVertex incidentEdge[2];
    {   /* b2FindIncidentEdge */
        const Vec2 normal1 = rmulT(xf2.q, rmul(xf1.q, poly1->n[edge1]));
        int index = 0;
        float minDot = b2_maxFloat;
        for (int i = 0; i < poly2->count; ++i) {
            const float dt = vdot(normal1, poly2->n[i]);
            if (dt < minDot) { minDot = dt; index = i; }
        }
        const int i1 = index, i2 = i1 + 1 < poly2->count ? i1 + 1 : 0;
        incidentEdge[0].v = xmul(xf2, poly2->v[i1]);
        incidentEdge[0].id.indexA = (uint8_t)edge1; incidentEdge[0].id.indexB = (uint8_t)i1;
        incidentEdge[0].id.typeA = CF_FACE; incidentEdge[0].id.typeB = CF_VERTEX;
        incidentEdge[1].v = xmul(xf2, poly2->v[i2]);
        incidentEdge[1].id.indexA = (uint8_t)edge1; incidentEdge[1].id.indexB = (uint8_t)i2;
        incidentEdge[1].id.typeA = CF_FACE; incidentEdge[1].id.typeB = CF_VERTEX;
    }
    const int count1 = poly1->count;
    const int iv1 = edge1, iv2 = edge1 + 1 < count1 ? edge1 + 1 : 0;
    Vec2 v11 = poly1->v[iv1], v12 = poly1->v[iv2];
    Vec2 localTangent = vsub(v12, v11);
    vnormalize(&localTangent);
    const Vec2 localNormal = vcross_vs(localTangent, 1.0f);
    const Vec2 planePoint = vscale(0.5f, vadd(v11, v12));
    const Vec2 tangent = rmul(xf1.q, localTangent);
    const Vec2 normal = vcross_vs(tangent, 1.0f);
    v11 = xmul(xf1, v11); v12 = xmul(xf1, v12);
    const float frontOffset = vdot(normal, v11);
    const float sideOffset1 = -vdot(tangent, v11) + totalRadius, sideOffset2 = vdot(tangent, v12) + totalRadius;
    ClipVertex clipPoints1[2], clipPoints2[2];
    if (clip_segment_to_line(clipPoints1, incidentEdge, vneg(tangent), sideOffset1, iv1) < 2) return;
    if (clip_segment_to_line(clipPoints2, clipPoints1, tangent, sideOffset2, iv2) < 2) return;
    m->localNormal = localNormal;
    m->localPoint = planePoint;
    int pointCount = 0;
    for (int i = 0; i < b2_maxManifoldPoints; ++i) {
        const float separation = vdot(normal, clipPoints2[i].v) - frontOffset;
        if (separation <= totalRadius) {
            ManifoldPoint *cp = &m->points[pointCount];
            cp->localPoint = xmulT(xf2, clipPoints2[i].v);
            cp->id = clipPoints2[i].id;
            if (flip) {
                const Feature cf = cp->id;
                cp->id.indexA = cf.indexB; cp->id.indexB = cf.indexA; cp->id.typeA = cf.typeB; cp->id.typeB = cf.typeA;
            }
            ++pointCount;
        }
    }
    m->pointCount = pointCount;
}

/* b2CollideEdge.cpp: b2EPCollider::Collide for an edge WITHOUT ghost vertices (m_hasVertex0 = m_hasVertex3 = false) */
static void collide_edge_and_polygon(Manifold *manifold, const Shape *edgeA, Xform xfA, const Shape *polygonB, Xform xfB) {
    const Xform xf = xmulT_xx(xfA, xfB);
    const Vec2 centroidB = xmul(xf, polygonB->centroid);
    const Vec2 v1 = edgeA->v[0], v2 = edgeA->v[1];
    Vec2 edge1 = vsub(v2, v1);
    vnormalize(&edge1);
    const Vec2 normal1 = V(edge1.y, -edge1.x);
    const float offset1 = vdot(normal1, vsub(centroidB, v1));
    const int front = offset1 >= 0.0f;
    Vec2 m_normal, lowerLimit, upperLimit;
    if (front) { m_normal = normal1; lowerLimit = vneg(normal1); upperLimit = vneg(normal1); }
    else { m_normal = vneg(normal1); lowerLimit = normal1; upperLimit = normal1; }
    Vec2 pv[b2_maxPolygonVertices], pn[b2_maxPolygonVertices];   /* polygonB in frame A */
    const int count = polygonB->count;
    for (int i = 0; i < count; ++i) { pv[i] = xmul(xf, polygonB->v[i]); pn[i] = rmul(xf.q, polygonB->n[i]); }
    const float radius = 2.0f * b2_polygonRadius;
    manifold->pointCount = 0;
    /* ComputeEdgeSeparation */
    float edgeSep = FLT_MAX;
    for (int i = 0; i < count; ++i) {
        const float s = vdot(m_normal, vsub(pv[i], v1));
        if (s < edgeSep) edgeSep = s;
    }
    if (edgeSep > radius) return;
    /* ComputePolygonSeparation */
    int polyType = 0 /* e_unknown */, polyIndex = -1;
    float polySep = -FLT_MAX;
    const Vec2 perp = V(-m_normal.y, m_normal.x);
    for (int i = 0; i < count; ++i) {
        const Vec2 n = vneg(pn[i]);
        const float s1 = vdot(n, vsub(pv[i], v1)), s2 = vdot(n, vsub(pv[i], v2));
        const float s = b2minf(s1, s2);
        if (s > radius) { polyType = 2; polyIndex = i; polySep = s; break; }   /* no collision */
        if (vdot(n, perp) >= 0.0f) { if (vdot(vsub(n, upperLimit), m_normal) < -b2_angularSlop) continue; }
        else { if (vdot(vsub(n, lowerLimit), m_normal) < -b2_angularSlop) continue; }
        if (s > polySep) { polyType = 2; polyIndex = i; polySep = s; }
    }
    if (polyType != 0 && polySep > radius) return;
    const float k_relativeTol = 0.98f, k_absoluteTol = 0.001f;
    int primaryIsEdge;
    if (polyType == 0) primaryIsEdge = 1;
    else if (polySep > k_relativeTol * edgeSep + k_absoluteTol) primaryIsEdge = 0;
    else primaryIsEdge = 1;
    ClipVertex ie[2];
    int rf_i1, rf_i2;
    Vec2 rf_v1, rf_v2, rf_normal;
    if (primaryIsEdge) {
        manifold->type = MF_FACE_A;
        int bestIndex = 0;
        float bestValue = vdot(m_normal, pn[0]);
        for (int i = 1; i < count; ++i) {
            const float value = vdot(m_normal, pn[i]);
            if (value < bestValue) { bestValue = value; bestIndex = i; }
        }
        const int i1 = bestIndex, i2 = i1 + 1 < count ? i1 + 1 : 0;
        ie[0].v = pv[i1]; ie[0].id.indexA = 0; ie[0].id.indexB = (uint8_t)i1; ie[0].id.typeA = CF_FACE; ie[0].id.typeB = CF_VERTEX;
        ie[1].v = pv[i2]; ie[1].id.indexA = 0; ie[1].id.indexB = (uint8_t)i2; ie[1].id.typeA = CF_FACE; ie[1].id.typeB = CF_VERTEX;
        if (front) { rf_i1 = 0; rf_i2 = 1; rf_v1 = v1; rf_v2 = v2; rf_normal = normal1; }
        else { rf_i1 = 1; rf_i2 = 0; rf_v1 = v2; rf_v2 = v1; rf_normal = vneg(normal1); }
    } else {
        manifold->type = MF_FACE_B;
        ie[0].v = v1; ie[0].id.indexA = 0; ie[0].id.indexB = (uint8_t)polyIndex; ie[0].id.typeA = CF_VERTEX; ie[0].id.typeB = CF_FACE;
        ie[1].v = v2; ie[1].id.indexA = 0; ie[1].id.indexB = (uint8_t)polyIndex; ie[1].id.typeA = CF_VERTEX; ie[1].id.typeB = CF_FACE;
        rf_i1 = polyIndex; rf_i2 = rf_i1 + 1 < count ? rf_i1 + 1 : 0;
        rf_v1 = pv[rf_i1]; rf_v2 = pv[rf_i2]; rf_normal = pn[rf_i1];
    }
    const Vec2 sideNormal1 = V(rf_normal.y, -rf_normal.x), sideNormal2 = vneg(sideNormal1);
    const float sideOffset1 = vdot(sideNormal1, rf_v1), sideOffset2 = vdot(sideNormal2, rf_v2);
    ClipVertex clipPoints1[2], clipPoints2[2];
    if (clip_segment_to_line(clipPoints1, ie, sideNormal1, sideOffset1, rf_i1) < b2_maxManifoldPoints) return;
    if (clip_segment_to_line(clipPoints2, clipPoints1, sideNormal2, sideOffset2, rf_i2) < b2_maxManifoldPoints) return;
    if (primaryIsEdge) { manifold->localNormal = rf_normal; manifold->localPoint = rf_v1; }
    else { manifold->localNormal = polygonB->n[rf_i1]; manifold->localPoint = polygonB->v[rf_i1]; }
    int pointCount = 0;
    for (int i = 0; i < b2_maxManifoldPoints; ++i) {
        const float separation = vdot(rf_normal, vsub(clipPoints2[i].v, rf_v1));
        if (separation <= radius) {
            ManifoldPoint *cp = &manifold->points[pointCount];
            if (primaryIsEdge) {
                cp->localPoint = xmulT(xf, clipPoints2[i].v);
                cp->id = clipPoints2[i].id;
            } else {
                cp->localPoint = clipPoints2[i].v;
                cp->id.typeA = clipPoints2[i].id.typeB; cp->id.typeB = clipPoints2[i].id.typeA;
                cp->id.indexA = clipPoints2[i].id.indexB; cp->id.indexB = clipPoints2[i].id.indexA;
            }
            ++pointCount;
        }
    }
    manifold->pointCount = pointCount;
}

/* b2WorldManifold::Initialize */
typedef struct { Vec2 normal, points[b2_maxManifoldPoints]; } WorldManifold;
static void world_manifold(WorldManifold *wm, const Manifold *m, Xform xfA, float radiusA, Xform xfB, float radiusB) {
    if (m->pointCount == 0) return;
    if (m->type == MF_FACE_A) {
        wm->normal = rmul(xfA.q, m->localNormal);
        const Vec2 planePoint = xmul(xfA, m->localPoint);
        for (int i = 0; i < m->pointCount; ++i) {
            const Vec2 clipPoint = xmul(xfB, m->points[i].localPoint);
            const Vec2 cA = vadd(clipPoint, vscale(radiusA - vdot(vsub(clipPoint, planePoint), wm->normal), wm->normal));
            const Vec2 cB = vsub(clipPoint, vscale(radiusB, wm->normal));
            wm->points[i] = vscale(0.5f, vadd(cA, cB));
        }
    } else {
        wm->normal = rmul(xfB.q, m->localNormal);
        const Vec2 planePoint = xmul(xfB, m->localPoint);
        for (int i = 0; i < m->pointCount; ++i) {
            const Vec2 clipPoint = xmul(xfA, m->points[i].localPoint);
            const Vec2 cB = vadd(clipPoint, vscale(radiusB - vdot(vsub(clipPoint, planePoint), wm->normal), wm->normal));
            const Vec2 cA = vsub(clipPoint, vscale(radiusA, wm->normal));
            wm->points[i] = vscale(0.5f, vadd(cA, cB));
        }
        wm->normal = vneg(wm->normal);   /* ensure the normal points from A to B */
    }
}

/* ------------------------------------------------------------------------------------------------ b2Distance.cpp (GJK) */
typedef struct { const Vec2 *v; int count; float radius; } DProxy;
static DProxy dproxy_of(const Shape *s) { DProxy p; p.v = s->v; p.count = s->type == SHAPE_EDGE ? 2 : s->count; p.radius = s->radius; return p; }
static int dproxy_support(const DProxy *p, Vec2 d) {
    int bestIndex = 0;
    float bestValue = vdot(p->v[0], d);
    for (int i = 1; i < p->count; ++i) {
        const float value = vdot(p->v[i], d);
        if (value > bestValue) { bestIndex = i; bestValue = value; }
    }
    return bestIndex;
}
typedef struct { float metric; uint16_t count; uint8_t indexA[3], indexB[3]; } SimplexCache;
typedef struct { Vec2 wA, wB, w; float a; int indexA, indexB; } SimplexVertex;
typedef struct { SimplexVertex v[3]; int count; } Simplex;

static float simplex_metric(const Simplex *s) {
    switch (s->count) {
        case 1: return 0.0f;
        case 2: return vlen(vsub(s->v[0].w, s->v[1].w));
        case 3: return vcross(vsub(s->v[1].w, s->v[0].w), vsub(s->v[2].w, s->v[0].w));
        default: return 0.0f;
    }
}
static void simplex_read_cache(Simplex *s, const SimplexCache *cache, const DProxy *pA, Xform xfA, const DProxy *pB, Xform xfB) {
    s->count = cache->count;
    for (int i = 0; i < s->count; ++i) {
        SimplexVertex *v = &s->v[i];
        v->indexA = cache->indexA[i]; v->indexB = cache->indexB[i];
        v->wA = xmul(xfA, pA->v[v->indexA]); v->wB = xmul(xfB, pB->v[v->indexB]);
        v->w = vsub(v->wB, v->wA);
        v->a = 0.0f;
    }
    if (s->count > 1) {   /* flush the cache if the metric changed too much */
        const float metric1 = cache->metric, metric2 = simplex_metric(s);
        if (metric2 < 0.5f * metric1 || 2.0f * metric1 < metric2 || metric2 < b2_epsilon) s->count = 0;
    }
    if (s->count == 0) {
        SimplexVertex *v = &s->v[0];
        v->indexA = 0; v->indexB = 0;
        v->wA = xmul(xfA, pA->v[0]); v->wB = xmul(xfB, pB->v[0]);
        v->w = vsub(v->wB, v->wA);
        v->a = 1.0f;
        s->count = 1;
    }
}
static void simplex_write_cache(const Simplex *s, SimplexCache *cache) {
    cache->metric = simplex_metric(s);
    cache->count = (uint16_t)s->count;
    for (int i = 0; i < s->count; ++i) { cache->indexA[i] = (uint8_t)s->v[i].indexA; cache->indexB[i] = (uint8_t)s->v[i].indexB; }
}
static void simplex_solve2(Simplex *s) {
    const Vec2 w1 = s->v[0].w, w2 = s->v[1].w, e12 = vsub(w2, w1);
    const float d12_2 = -vdot(w1, e12);
    if (d12_2 <= 0.0f) { s->v[0].a = 1.0f; s->count = 1; return; }
    const float d12_1 = vdot(w2, e12);
    if (d12_1 <= 0.0f) { s->v[1].a = 1.0f; s->count = 1; s->v[0] = s->v[1]; return; }
    const float inv_d12 = 1.0f / (d12_1 + d12_2);
    s->v[0].a = d12_1 * inv_d12; s->v[1].a = d12_2 * inv_d12; s->count = 2;
}
static void simplex_solve3(Simplex *s) {
    const Vec2 w1 = s->v[0].w, w2 = s->v[1].w, w3 = s->v[2].w;
    const Vec2 e12 = vsub(w2, w1);
    const float w1e12 = vdot(w1, e12), w2e12 = vdot(w2, e12);
    const float d12_1 = w2e12, d12_2 = -w1e12;
    const Vec2 e13 = vsub(w3, w1);
    const float w1e13 = vdot(w1, e13), w3e13 = vdot(w3, e13);
    const float d13_1 = w3e13, d13_2 = -w1e13;
    const Vec2 e23 = vsub(w3, w2);
    const float w2e23 = vdot(w2, e23), w3e23 = vdot(w3, e23);
    const float d23_1 = w3e23, d23_2 = -w2e23;
    const float n123 = vcross(e12, e13);
    const float d123_1 = n123 * vcross(w2, w3), d123_2 = n123 * vcross(w3, w1), d123_3 = n123 * vcross(w1, w2);
    if (d12_2 <= 0.0f && d13_2 <= 0.0f) { s->v[0].a = 1.0f; s->count = 1; return; }
    if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f) {
        const float inv = 1.0f / (d12_1 + d12_2);
        s->v[0].a = d12_1 * inv; s->v[1].a = d12_2 * inv; s->count = 2; return;
    }
    if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f) {
        const float inv = 1.0f / (d13_1 + d13_2);
        s->v[0].a = d13_1 * inv; s->v[2].a = d13_2 * inv; s->count = 2; s->v[1] = s->v[2]; return;
    }
    if (d12_1 <= 0.0f && d23_2 <= 0.0f) { s->v[1].a = 1.0f; s->count = 1; s->v[0] = s->v[1]; return; }
    if (d13_1 <= 0.0f && d23_1 <= 0.0f) { s->v[2].a = 1.0f; s->count = 1; s->v[0] = s->v[2]; return; }
    if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f) {
        const float inv = 1.0f / (d23_1 + d23_2);
        s->v[1].a = d23_1 * inv; s->v[2].a = d23_2 * inv; s->count = 2; s->v[0] = s->v[2]; return;
    }
    const float inv = 1.0f / (d123_1 + d123_2 + d123_3);
    s->v[0].a = d123_1 * inv; s->v[1].a = d123_2 * inv; s->v[2].a = d123_3 * inv; s->count = 3;
}
/* b2Distance with useRadii = false; returns the distance between the core shapes and updates the cache */
static float gjk_distance(SimplexCache *cache, const DProxy *pA, Xform xfA, const DProxy *pB, Xform xfB) {
    Simplex simplex;
    simplex_read_cache(&simplex, cache, pA, xfA, pB, xfB);
    const int k_maxIters = 20;
    int saveA[3], saveB[3], saveCount = 0;
    int iter = 0;
    while (iter < k_maxIters) {
        saveCount = simplex.count;
        for (int i = 0; i < saveCount; ++i) { saveA[i] = simplex.v[i].indexA; saveB[i] = simplex.v[i].indexB; }
        switch (simplex.count) {
            case 1: break;
            case 2: simplex_solve2(&simplex); break;
            case 3: simplex_solve3(&simplex); break;
        }
        if (simplex.count == 3) break;   /* the origin is inside the triangle: overlap */
        Vec2 d;                          /* GetSearchDirection */
        if (simplex.count == 1) d = vneg(simplex.v[0].w);
        else {
            const Vec2 e12 = vsub(simplex.v[1].w, simplex.v[0].w);
            const float sgn = vcross(e12, vneg(simplex.v[0].w));
            d = sgn > 0.0f ? vcross_sv(1.0f, e12) : vcross_vs(e12, 1.0f);
        }
        if (vlen2(d) < b2_epsilon * b2_epsilon) break;   /* the origin is probably contained by a line segment or triangle */
        SimplexVertex *vertex = &simplex.v[simplex.count];
        vertex->indexA = dproxy_support(pA, rmulT(xfA.q, vneg(d)));
        vertex->wA = xmul(xfA, pA->v[vertex->indexA]);
        vertex->indexB = dproxy_support(pB, rmulT(xfB.q, d));
        vertex->wB = xmul(xfB, pB->v[vertex->indexB]);
        vertex->w = vsub(vertex->wB, vertex->wA);
        ++iter;
        int duplicate = 0;
        for (int i = 0; i < saveCount; ++i)
            if (vertex->indexA == saveA[i] && vertex->indexB == saveB[i]) { duplicate = 1; break; }
        if (duplicate) break;
        ++simplex.count;
    }
    Vec2 pointA, pointB;   /* GetWitnessPoints */
    if (simplex.count == 1) { pointA = simplex.v[0].wA; pointB = simplex.v[0].wB; }
    else if (simplex.count == 2) {
        pointA = vadd(vscale(simplex.v[0].a, simplex.v[0].wA), vscale(simplex.v[1].a, simplex.v[1].wA));
        pointB = vadd(vscale(simplex.v[0].a, simplex.v[0].wB), vscale(simplex.v[1].a, simplex.v[1].wB));
    } else {
        pointA = vadd(vadd(vscale(simplex.v[0].a, simplex.v[0].wA), vscale(simplex.v[1].a, simplex.v[1].wA)), vscale(simplex.v[2].a, simplex.v[2].wA));
        pointB = pointA;
    }
    simplex_write_cache(&simplex, cache);
    return vlen(vsub(pointA, pointB));
}

/* ------------------------------------------------------------------------------------------------ b2TimeOfImpact.cpp */
enum { SEP_POINTS = 0, SEP_FACE_A = 1, SEP_FACE_B = 2 };
typedef struct { const DProxy *pA, *pB; Sweep sweepA, sweepB; int type; Vec2 localPoint, axis; } SepFn;
static float sepfn_initialize(SepFn *f, const SimplexCache *cache, const DProxy *pA, const Sweep *sA, const DProxy *pB, const Sweep *sB, float t1) {
    f->pA = pA; f->pB = pB; f->sweepA = *sA; f->sweepB = *sB;
    const int count = cache->count;
    const Xform xfA = sweep_transform(sA, t1), xfB = sweep_transform(sB, t1);
    if (count == 1) {
        f->type = SEP_POINTS;
        const Vec2 pointA = xmul(xfA, pA->v[cache->indexA[0]]), pointB = xmul(xfB, pB->v[cache->indexB[0]]);
        f->axis = vsub(pointB, pointA);
        return vnormalize(&f->axis);
    } else if (cache->indexA[0] == cache->indexA[1]) {   /* two points on B and one on A */
        f->type = SEP_FACE_B;
        const Vec2 localPointB1 = pB->v[cache->indexB[0]], localPointB2 = pB->v[cache->indexB[1]];
        f->axis = vcross_vs(vsub(localPointB2, localPointB1), 1.0f);
        vnormalize(&f->axis);
        const Vec2 normal = rmul(xfB.q, f->axis);
        f->localPoint = vscale(0.5f, vadd(localPointB1, localPointB2));
        const Vec2 pointB = xmul(xfB, f->localPoint), pointA = xmul(xfA, pA->v[cache->indexA[0]]);
        float s = vdot(vsub(pointA, pointB), normal);
        if (s < 0.0f) { f->axis = vneg(f->axis); s = -s; }
        return s;
    } else {                                             /* two points on A and one or two points on B */
        f->type = SEP_FACE_A;
        const Vec2 localPointA1 = pA->v[cache->indexA[0]], localPointA2 = pA->v[cache->indexA[1]];
        f->axis = vcross_vs(vsub(localPointA2, localPointA1), 1.0f);
        vnormalize(&f->axis);
        const Vec2 normal = rmul(xfA.q, f->axis);
        f->localPoint = vscale(0.5f, vadd(localPointA1, localPointA2));
        const Vec2 pointA = xmul(xfA, f->localPoint), pointB = xmul(xfB, pB->v[cache->indexB[0]]);
        float s = vdot(vsub(pointB, pointA), normal);
        if (s < 0.0f) { f->axis = vneg(f->axis); s = -s; }
        return s;
    }
}
static float sepfn_find_min(const SepFn *f, int *indexA, int *indexB, float t) {
    const Xform xfA = sweep_transform(&f->sweepA, t), xfB = sweep_transform(&f->sweepB, t);
    if (f->type == SEP_POINTS) {
        const Vec2 axisA = rmulT(xfA.q, f->axis), axisB = rmulT(xfB.q, vneg(f->axis));
        *indexA = dproxy_support(f->pA, axisA); *indexB = dproxy_support(f->pB, axisB);
        const Vec2 pointA = xmul(xfA, f->pA->v[*indexA]), pointB = xmul(xfB, f->pB->v[*indexB]);
        return vdot(vsub(pointB, pointA), f->axis);
    } else if (f->type == SEP_FACE_A) {
        const Vec2 normal = rmul(xfA.q, f->axis), pointA = xmul(xfA, f->localPoint);
        const Vec2 axisB = rmulT(xfB.q, vneg(normal));
        *indexA = -1; *indexB = dproxy_support(f->pB, axisB);
        const Vec2 pointB = xmul(xfB, f->pB->v[*indexB]);
        return vdot(vsub(pointB, pointA), normal);
    } else {
        const Vec2 normal = rmul(xfB.q, f->axis), pointB = xmul(xfB, f->localPoint);
        const Vec2 axisA = rmulT(xfA.q, vneg(normal));
        *indexB = -1; *indexA = dproxy_support(f->pA, axisA);
        const Vec2 pointA = xmul(xfA, f->pA->v[*indexA]);
        return vdot(vsub(pointA, pointB), normal);
    }
}
static float sepfn_evaluate(const SepFn *f, int indexA, int indexB, float t) {
    const Xform xfA = sweep_transform(&f->sweepA, t), xfB = sweep_transform(&f->sweepB, t);
    if (f->type == SEP_POINTS) {
        const Vec2 pointA = xmul(xfA, f->pA->v[indexA]), pointB = xmul(xfB, f->pB->v[indexB]);
        return vdot(vsub(pointB, pointA), f->axis);
    } else if (f->type == SEP_FACE_A) {
        const Vec2 normal = rmul(xfA.q, f->axis), pointA = xmul(xfA, f->localPoint), pointB = xmul(xfB, f->pB->v[indexB]);
        return vdot(vsub(pointB, pointA), normal);
    } else {
        const Vec2 normal = rmul(xfB.q, f->axis), pointB = xmul(xfB, f->localPoint), pointA = xmul(xfA, f->pA->v[indexA]);
        return vdot(vsub(pointA, pointB), normal);
    }
}
enum { TOI_UNKNOWN = 0, TOI_FAILED, TOI_OVERLAPPED, TOI_TOUCHING, TOI_SEPARATED };
static int time_of_impact(float *t_out, const DProxy *pA, const Sweep *sweepA_in, const DProxy *pB, const Sweep *sweepB_in, float tMax) {
    int state = TOI_UNKNOWN;
    *t_out = tMax;
    Sweep sweepA = *sweepA_in, sweepB = *sweepB_in;
    sweep_normalize(&sweepA); sweep_normalize(&sweepB);   /* large rotations can make the root finder fail */
    const float totalRadius = pA->radius + pB->radius;
    const float target = b2maxf(b2_linearSlop, totalRadius - 3.0f * b2_linearSlop);
    const float tolerance = 0.25f * b2_linearSlop;
    float t1 = 0.0f;
    const int k_maxIterations = 20;
    int iter = 0;
    SimplexCache cache;
    memset(&cache, 0, sizeof(cache));
    for (;;) {
        const Xform xfA = sweep_transform(&sweepA, t1), xfB = sweep_transform(&sweepB, t1);
        const float distance = gjk_distance(&cache, pA, xfA, pB, xfB);
        if (distance <= 0.0f) { state = TOI_OVERLAPPED; *t_out = 0.0f; break; }           /* failure */
        if (distance < target + tolerance) { state = TOI_TOUCHING; *t_out = t1; break; }  /* victory */
        SepFn fcn;
        sepfn_initialize(&fcn, &cache, pA, &sweepA, pB, &sweepB, t1);
        int done = 0;
        float t2 = tMax;
        int pushBackIter = 0;
        for (;;) {
            int indexA, indexB;
            float s2 = sepfn_find_min(&fcn, &indexA, &indexB, t2);
            if (s2 > target + tolerance) { state = TOI_SEPARATED; *t_out = tMax; done = 1; break; }
            if (s2 > target - tolerance) { t1 = t2; break; }   /* advance the sweeps */
            float s1 = sepfn_evaluate(&fcn, indexA, indexB, t1);
            if (s1 < target - tolerance) { state = TOI_FAILED; *t_out = t1; done = 1; break; }
            if (s1 <= target + tolerance) { state = TOI_TOUCHING; *t_out = t1; done = 1; break; }
            int rootIterCount = 0;
            float a1 = t1, a2 = t2;
            for (;;) {   /* 1D root of f(t) - target: secant and bisection alternate */
                float t;
                if (rootIterCount & 1) t = a1 + (target - s1) * (a2 - a1) / (s2 - s1);
                else t = 0.5f * (a1 + a2);
                ++rootIterCount;
                const float s = sepfn_evaluate(&fcn, indexA, indexB, t);
                if (fabsf(s - target) < tolerance) { t2 = t; break; }
                if (s > target) { a1 = t; s1 = s; } else { a2 = t; s2 = s; }
                if (rootIterCount == 50) break;
            }
            ++pushBackIter;
            if (pushBackIter == b2_maxPolygonVertices) break;
        }
        ++iter;
        if (done) break;
        if (iter == k_maxIterations) { state = TOI_FAILED; *t_out = t1; break; }   /* root finder got stuck */
    }
    return state;
}

/* ------------------------------------------------------------------------------------------------ world data (b2Body, b2Contact, b2Joint) */
#define MWR_MAX_BODIES 320      /* 1 package + 5 x 10 walker bodies + (TERRAIN_LENGTH * 10 / 8 - 1) terrain edges: the reference's curriculum runs 2 .. 10 walkers (lessons/multiwalker/env.yaml) */
#define MWR_MAX_CONTACTS 2048
#define MWR_MAX_JOINTS 40
enum { BODY_STATIC = 0, BODY_DYNAMIC = 2 };
enum { LIMIT_INACTIVE = 0, LIMIT_AT_LOWER = 1, LIMIT_AT_UPPER = 2, LIMIT_EQUAL = 3 };

typedef struct {
    int type;
    Xform xf;
    Sweep sweep;
    Vec2 linearVelocity; float angularVelocity;
    Vec2 force; float torque;
    float mass, invMass, I, invI;
    float sleepTime;
    int awake, islandFlag, islandIndex;
    /* the body's single fixture */
    Shape shape;
    float density, friction;
    uint16_t categoryBits, maskBits;
    AABB fatAABB;          /* b2DynamicTree node AABB of the fixture's proxy */
    int proxyId;           /* creation order (D1) */
    int moved;             /* in the broad-phase move buffer */
    int contactList;       /* head of the contact-edge list: edge = contact * 2 + side, -1 = empty */
    int jointList;         /* head of the joint-edge list:   edge = joint * 2 + side */
    int prev, next;        /* world body list */
    int userKind, userIndex, userFlag;   /* env bookkeeping (ContactDetector): kind, walker, lower leg's ground_contact */
} Body;

typedef struct {
    int used;
    int bodyA, bodyB;      /* = fixture A / B (one fixture per body) */
    Manifold manifold;
    int touching, enabled, islandFlag, toiFlag, filterFlag;
    int toiCount;
    float toi, friction, restitution;
    int prev, next;        /* world contact list */
    int edgePrev[2], edgeNext[2];   /* m_nodeA / m_nodeB in the bodies' contact-edge lists */
} Contact;

typedef struct {
    int bodyA, bodyB;
    Vec2 localAnchorA, localAnchorB;
    float referenceAngle, lowerAngle, upperAngle, maxMotorTorque, motorSpeed;
    int enableLimit, enableMotor, collideConnected;
    float impulse[3], motorImpulse;   /* m_impulse (b2Vec3), m_motorImpulse */
    int limitState;
    int islandFlag;
    int prev, next;
    int edgePrev[2], edgeNext[2];
    /* solver temp */
    int indexA, indexB;
    Vec2 rA, rB, localCenterA, localCenterB;
    float invMassA, invMassB, invIA, invIB;
    float mass[9];         /* b2Mat33 m_mass: ex = [0..2], ey = [3..5], ez = [6..8] */
    float motorMass;
} RevoluteJoint;

struct World;
typedef void (*ContactCallback)(void *user, struct World *w, int contact, int begin);

typedef struct World {
    Body bodies[MWR_MAX_BODIES];
    int bodyCount, bodyList;
    Contact contacts[MWR_MAX_CONTACTS];
    int contactList, contactCount, contactFree;   /* contactFree: lowest never-used slot (slots of destroyed contacts are recycled first) */
    RevoluteJoint joints[MWR_MAX_JOINTS];
    int jointCount, jointList;
    Vec2 gravity;
    int allowSleep, continuousPhysics, warmStarting, newFixture, stepComplete;
    int polygonRevision;   /* 0: b2CollidePolygons of 2.3.0 (hill-climbing b2FindMaxSeparation), 1: of later 2.3.x revisions (see collide_polygons) */
    float inv_dt0;
    ContactCallback listener;
    void *listenerUser;
    int proxyCount;
    long stat_toi_events, stat_contacts_created;
    long stat_rays, stat_rays_hit, stat_rays_multi;   /* lidar rays cast / with a hit / with hits on SEVERAL edges (where D2's "closest" and Box2D's "first in tree order" can differ) */
} World;

static void world_init(World *w, Vec2 gravity) {
    memset(w, 0, sizeof(*w));
    w->bodyList = -1; w->contactList = -1; w->jointList = -1;
    w->gravity = gravity;
    w->allowSleep = 1; w->continuousPhysics = 1; w->warmStarting = 1; w->stepComplete = 1;
    for (int i = 0; i < MWR_MAX_CONTACTS; ++i) w->contacts[i].used = 0;
}

static void body_set_awake(Body *b, int flag) {            /* b2Body::SetAwake */
    if (flag) {
        if (!b->awake) { b->awake = 1; b->sleepTime = 0.0f; }
    } else {
        b->awake = 0; b->sleepTime = 0.0f;
        b->linearVelocity = V(0, 0); b->angularVelocity = 0.0f; b->force = V(0, 0); b->torque = 0.0f;
    }
}
static void body_sync_transform(Body *b) {                 /* b2Body::SynchronizeTransform */
    b->xf.q = rot_of(b->sweep.a);
    b->xf.p = vsub(b->sweep.c, rmul(b->xf.q, b->sweep.localCenter));
}
static void body_advance(Body *b, float alpha) {           /* b2Body::Advance */
    sweep_advance(&b->sweep, alpha);
    b->sweep.c = b->sweep.c0; b->sweep.a = b->sweep.a0;
    b->xf.q = rot_of(b->sweep.a);
    b->xf.p = vsub(b->sweep.c, rmul(b->xf.q, b->sweep.localCenter));
}

/* b2World::CreateBody + b2Body::CreateFixture (+ ResetMassData): one fixture per body.  New bodies go to the FRONT of the list. */
static int world_create_body(World *w, int type, Vec2 position, float angle, const Shape *shape, float density, float friction,
                             uint16_t categoryBits, uint16_t maskBits) {
    const int id = w->bodyCount++;
    Body *b = &w->bodies[id];
    memset(b, 0, sizeof(*b));
    b->type = type;
    b->xf.p = position; b->xf.q = rot_of(angle);
    b->sweep.localCenter = V(0, 0);
    b->sweep.c0 = b->xf.p; b->sweep.c = b->xf.p; b->sweep.a0 = angle; b->sweep.a = angle; b->sweep.alpha0 = 0.0f;
    b->awake = 1;
    b->contactList = -1; b->jointList = -1;
    b->prev = -1; b->next = w->bodyList;
    if (w->bodyList >= 0) w->bodies[w->bodyList].prev = id;
    w->bodyList = id;
    b->shape = *shape; b->density = density; b->friction = friction; b->categoryBits = categoryBits; b->maskBits = maskBits;
    /* b2Fixture::CreateProxies: tight AABB at the body transform, fattened by b2_aabbExtension; the proxy is buffered as moved */
    const AABB aabb = shape_aabb(&b->shape, b->xf);
    b->fatAABB.lo = V(aabb.lo.x - b2_aabbExtension, aabb.lo.y - b2_aabbExtension);
    b->fatAABB.hi = V(aabb.hi.x + b2_aabbExtension, aabb.hi.y + b2_aabbExtension);
    /* D1: creation order, as in a fresh b2World.  MWR_PROXY_ORDER=reverse in the environment (a sensitivity probe of the test infrastructure, DESIGN.md
     * section 2) hands the ids out in DESCENDING order instead: the most different order a recycled b2DynamicTree node pool could produce */
    {
        static int reverse = -1;
        if (reverse < 0) { const char *e = getenv("MWR_PROXY_ORDER"); reverse = (e && e[0] == 'r') ? 1 : 0; }
        b->proxyId = reverse ? MWR_MAX_BODIES - 1 - w->proxyCount : w->proxyCount;
        ++w->proxyCount;
    }
    b->moved = 1;
    w->newFixture = 1;
    /* b2Body::ResetMassData */
    if (type == BODY_DYNAMIC && density > 0.0f) {
        const MassData md = polygon_mass(&b->shape, density);
        b->mass = md.mass;
        Vec2 localCenter = vscale(md.mass, md.center);
        b->I = md.I;
        if (b->mass > 0.0f) { b->invMass = 1.0f / b->mass; localCenter = vscale(b->invMass, localCenter); }
        else { b->mass = 1.0f; b->invMass = 1.0f; }
        if (b->I > 0.0f) { b->I -= b->mass * vdot(localCenter, localCenter); b->invI = 1.0f / b->I; }
        else { b->I = 0.0f; b->invI = 0.0f; }
        const Vec2 oldCenter = b->sweep.c;
        b->sweep.localCenter = localCenter;
        b->sweep.c0 = b->sweep.c = xmul(b->xf, b->sweep.localCenter);
        b->linearVelocity = vadd(b->linearVelocity, vcross_sv(b->angularVelocity, vsub(b->sweep.c, oldCenter)));
    }
    return id;
}

/* b2World::CreateJoint(b2RevoluteJointDef): new joints go to the FRONT of the world list and of both bodies' edge lists */
static int world_create_revolute(World *w, int bodyA, int bodyB, Vec2 anchorA, Vec2 anchorB, float lower, float upper,
                                 float maxMotorTorque, float motorSpeed) {
    const int id = w->jointCount++;
    RevoluteJoint *j = &w->joints[id];
    memset(j, 0, sizeof(*j));
    j->bodyA = bodyA; j->bodyB = bodyB; j->localAnchorA = anchorA; j->localAnchorB = anchorB;
    j->referenceAngle = 0.0f;   /* the def is built from keyword arguments (:145-157), not b2RevoluteJointDef::Initialize */
    j->lowerAngle = lower; j->upperAngle = upper; j->maxMotorTorque = maxMotorTorque; j->motorSpeed = motorSpeed;
    j->enableLimit = 1; j->enableMotor = 1; j->collideConnected = 0;
    j->limitState = LIMIT_INACTIVE;
    j->prev = -1; j->next = w->jointList;
    if (w->jointList >= 0) w->joints[w->jointList].prev = id;
    w->jointList = id;
    const int body[2] = {bodyA, bodyB};
    for (int side = 0; side < 2; ++side) {
        Body *b = &w->bodies[body[side]];
        j->edgePrev[side] = -1; j->edgeNext[side] = b->jointList;
        if (b->jointList >= 0) w->joints[b->jointList >> 1].edgePrev[b->jointList & 1] = id * 2 + side;
        b->jointList = id * 2 + side;
    }
    /* collideConnected == false: contacts between the two bodies would be flagged for filtering; none exist at creation */
    return id;
}

/* ------------------------------------------------------------------------------------------------ b2Contact / b2ContactManager */
static void contact_evaluate(World *w, Contact *c, Manifold *m) {
    Body *bA = &w->bodies[c->bodyA], *bB = &w->bodies[c->bodyB];
    if (bA->shape.type == SHAPE_EDGE) collide_edge_and_polygon(m, &bA->shape, bA->xf, &bB->shape, bB->xf);
    else collide_polygons(m, &bA->shape, bA->xf, &bB->shape, bB->xf, w->polygonRevision);
}
static void contact_update(World *w, int ci) {             /* b2Contact::Update */
    Contact *c = &w->contacts[ci];
    const Manifold oldManifold = c->manifold;
    c->enabled = 1;                                        /* re-enable this contact */
    const int wasTouching = c->touching;
    Body *bA = &w->bodies[c->bodyA], *bB = &w->bodies[c->bodyB];
    contact_evaluate(w, c, &c->manifold);
    const int touching = c->manifold.pointCount > 0;
    for (int i = 0; i < c->manifold.pointCount; ++i) {     /* match old contact ids to new ones, copy the stored impulses */
        ManifoldPoint *mp2 = &c->manifold.points[i];
        mp2->normalImpulse = 0.0f; mp2->tangentImpulse = 0.0f;
        const uint32_t id2 = feature_key(mp2->id);
        for (int j = 0; j < oldManifold.pointCount; ++j) {
            const ManifoldPoint *mp1 = &oldManifold.points[j];
            if (feature_key(mp1->id) == id2) { mp2->normalImpulse = mp1->normalImpulse; mp2->tangentImpulse = mp1->tangentImpulse; break; }
        }
    }
    if (touching != wasTouching) { body_set_awake(bA, 1); body_set_awake(bB, 1); }
    c->touching = touching;
    if (!wasTouching && touching && w->listener) w->listener(w->listenerUser, w, ci, 1);
    if (wasTouching && !touching && w->listener) w->listener(w->listenerUser, w, ci, 0);
}

static int body_should_collide(const World *w, const Body *b, int bi, int other) {   /* b2Body::ShouldCollide */
    if (b->type != BODY_DYNAMIC && w->bodies[other].type != BODY_DYNAMIC) return 0;  /* at least one body should be dynamic */
    for (int e = b->jointList; e >= 0; e = w->joints[e >> 1].edgeNext[e & 1]) {
        const RevoluteJoint *j = &w->joints[e >> 1];
        const int o = (e & 1) ? j->bodyA : j->bodyB;
        if (o == other && !j->collideConnected) return 0;
    }
    (void)bi;
    return 1;
}

static void contact_manager_add_pair(World *w, int bodyA, int bodyB) {   /* b2ContactManager::AddPair; proxyId(bodyA) < proxyId(bodyB) */
    if (bodyA == bodyB) return;
    Body *bB = &w->bodies[bodyB];
    for (int e = bB->contactList; e >= 0; e = w->contacts[e >> 1].edgeNext[e & 1]) {   /* does a contact already exist? */
        const Contact *c = &w->contacts[e >> 1];
        const int other = (e & 1) ? c->bodyA : c->bodyB;
        if (other == bodyA) return;   /* one fixture per body: same body pair = same fixture pair */
    }
    if (!body_should_collide(w, bB, bodyB, bodyA)) return;
    const Body *bA = &w->bodies[bodyA];
    /* b2ContactFilter::ShouldCollide (groupIndex is 0 everywhere) */
    if (!((bA->maskBits & bB->categoryBits) != 0 && (bA->categoryBits & bB->maskBits) != 0)) return;
    /* b2Contact::Create: edge + polygon pairs are stored edge first (s_registers[e_polygon][e_edge].primary == false) */
    int fA = bodyA, fB = bodyB;
    if (bA->shape.type == SHAPE_POLYGON && bB->shape.type == SHAPE_EDGE) { fA = bodyB; fB = bodyA; }
    if (w->bodies[fA].shape.type == SHAPE_EDGE && w->bodies[fB].shape.type == SHAPE_EDGE) return;   /* no edge-edge contact type */
    int ci = -1;
    for (int i = 0; i < MWR_MAX_CONTACTS; ++i) if (!w->contacts[i].used) { ci = i; break; }
    if (ci < 0) return;   /* pool exhausted (never with this scene: < 200 candidate pairs) */
    Contact *c = &w->contacts[ci];
    memset(c, 0, sizeof(*c));
    c->used = 1; c->enabled = 1;
    c->bodyA = fA; c->bodyB = fB;
    c->friction = sqrtf(w->bodies[fA].friction * w->bodies[fB].friction);   /* b2MixFriction */
    c->restitution = 0.0f;                                                   /* b2MixRestitution = max(0, 0) */
    c->toi = 1.0f;
    c->prev = -1; c->next = w->contactList;                                  /* insert into the world, at the front */
    if (w->contactList >= 0) w->contacts[w->contactList].prev = ci;
    w->contactList = ci;
    const int body[2] = {fA, fB};
    for (int side = 0; side < 2; ++side) {                                   /* connect to the island graph, at the front */
        Body *b = &w->bodies[body[side]];
        c->edgePrev[side] = -1; c->edgeNext[side] = b->contactList;
        if (b->contactList >= 0) w->contacts[b->contactList >> 1].edgePrev[b->contactList & 1] = ci * 2 + side;
        b->contactList = ci * 2 + side;
    }
    body_set_awake(&w->bodies[fA], 1); body_set_awake(&w->bodies[fB], 1);    /* wake up the bodies */
    ++w->contactCount; ++w->stat_contacts_created;
}

static void contact_manager_destroy(World *w, int ci) {    /* b2ContactManager::Destroy */
    Contact *c = &w->contacts[ci];
    if (w->listener && c->touching) w->listener(w->listenerUser, w, ci, 0);
    if (c->prev >= 0) w->contacts[c->prev].next = c->next;
    if (c->next >= 0) w->contacts[c->next].prev = c->prev;
    if (ci == w->contactList) w->contactList = c->next;
    const int body[2] = {c->bodyA, c->bodyB};
    for (int side = 0; side < 2; ++side) {
        Body *b = &w->bodies[body[side]];
        const int ep = c->edgePrev[side], en = c->edgeNext[side];
        if (ep >= 0) w->contacts[ep >> 1].edgeNext[ep & 1] = en;
        if (en >= 0) w->contacts[en >> 1].edgePrev[en & 1] = ep;
        if (b->contactList == ci * 2 + side) b->contactList = en;
    }
    if (c->manifold.pointCount > 0) { body_set_awake(&w->bodies[c->bodyA], 1); body_set_awake(&w->bodies[c->bodyB], 1); }   /* b2Contact::Destroy */
    c->used = 0;
    --w->contactCount;
}

static void contact_manager_collide(World *w) {            /* b2ContactManager::Collide */
    int ci = w->contactList;
    while (ci >= 0) {
        Contact *c = &w->contacts[ci];
        const int next = c->next;
        Body *bA = &w->bodies[c->bodyA], *bB = &w->bodies[c->bodyB];
        /* (e_filterFlag: set only by joint creation / filter changes after a contact exists; never here) */
        const int activeA = bA->awake && bA->type != BODY_STATIC, activeB = bB->awake && bB->type != BODY_STATIC;
        if (!activeA && !activeB) { ci = next; continue; }            /* at least one body must be awake and dynamic */
        if (!aabb_overlap(&bA->fatAABB, &bB->fatAABB)) {              /* here we destroy contacts that cease to overlap in the broad-phase */
            contact_manager_destroy(w, ci);
            ci = next;
            continue;
        }
        contact_update(w, ci);                                        /* the contact persists */
        ci = next;
    }
}

typedef struct { int a, b; } ProxyPair;
static int pair_less(const void *x, const void *y) {       /* b2PairLessThan */
    const ProxyPair *p = (const ProxyPair *)x, *q = (const ProxyPair *)y;
    if (p->a != q->a) return p->a < q->a ? -1 : 1;
    if (p->b != q->b) return p->b < q->b ? -1 : 1;
    return 0;
}
static void contact_manager_find_new_contacts(World *w) {  /* b2BroadPhase::UpdatePairs(b2ContactManager*) */
    static __thread ProxyPair pairs[MWR_MAX_BODIES * MWR_MAX_BODIES / 2];
    static __thread int proxyBody[MWR_MAX_BODIES];
    int pairCount = 0;
    for (int i = 0; i < w->bodyCount; ++i) proxyBody[w->bodies[i].proxyId] = i;
    for (int i = 0; i < w->bodyCount; ++i) {               /* every proxy in the move buffer queries the tree with its fat AABB */
        const Body *bi = &w->bodies[i];
        if (!bi->moved) continue;
        for (int k = 0; k < w->bodyCount; ++k) {
            if (k == i) continue;                           /* a proxy cannot form a pair with itself */
            const Body *bk = &w->bodies[k];
            if (!aabb_overlap(&bi->fatAABB, &bk->fatAABB)) continue;
            ProxyPair p;
            p.a = bi->proxyId < bk->proxyId ? bi->proxyId : bk->proxyId;
            p.b = bi->proxyId < bk->proxyId ? bk->proxyId : bi->proxyId;
            pairs[pairCount++] = p;
        }
    }
    for (int i = 0; i < w->bodyCount; ++i) w->bodies[i].moved = 0;   /* reset the move buffer */
    qsort(pairs, (size_t)pairCount, sizeof(ProxyPair), pair_less);   /* sort the pair buffer to expose duplicates */
    int i = 0;
    while (i < pairCount) {                                 /* send the pairs back to the client */
        const ProxyPair primary = pairs[i];
        contact_manager_add_pair(w, proxyBody[primary.a], proxyBody[primary.b]);
        ++i;
        while (i < pairCount && pairs[i].a == primary.a && pairs[i].b == primary.b) ++i;   /* skip any duplicate pairs */
    }
}

/* b2Body::SynchronizeFixtures -> b2Fixture::Synchronize -> b2BroadPhase::MoveProxy -> b2DynamicTree::MoveProxy */
static void body_sync_fixtures(World *w, Body *b) {
    (void)w;
    Xform xf1;
    xf1.q = rot_of(b->sweep.a0);
    xf1.p = vsub(b->sweep.c0, rmul(xf1.q, b->sweep.localCenter));
    const AABB aabb1 = shape_aabb(&b->shape, xf1), aabb2 = shape_aabb(&b->shape, b->xf);
    AABB aabb;   /* covers the swept shape (may miss some rotation effect) */
    aabb.lo = V(b2minf(aabb1.lo.x, aabb2.lo.x), b2minf(aabb1.lo.y, aabb2.lo.y));
    aabb.hi = V(b2maxf(aabb1.hi.x, aabb2.hi.x), b2maxf(aabb1.hi.y, aabb2.hi.y));
    const Vec2 displacement = vsub(b->xf.p, xf1.p);
    if (aabb_contains(&b->fatAABB, &aabb)) return;          /* still inside the fat AABB: nothing to do */
    AABB fat;
    fat.lo = V(aabb.lo.x - b2_aabbExtension, aabb.lo.y - b2_aabbExtension);
    fat.hi = V(aabb.hi.x + b2_aabbExtension, aabb.hi.y + b2_aabbExtension);
    const Vec2 d = vscale(b2_aabbMultiplier, displacement);  /* predict AABB displacement */
    if (d.x < 0.0f) fat.lo.x += d.x; else fat.hi.x += d.x;
    if (d.y < 0.0f) fat.lo.y += d.y; else fat.hi.y += d.y;
    b->fatAABB = fat;
    b->moved = 1;                                            /* b2BroadPhase::BufferMove */
}

/* ------------------------------------------------------------------------------------------------ b2ContactSolver */
typedef struct { Vec2 c; float a; } Position;
typedef struct { Vec2 v; float w; } Velocity;
typedef struct { Vec2 rA, rB; float normalImpulse, tangentImpulse, normalMass, tangentMass, velocityBias; } VCPoint;
typedef struct {
    VCPoint points[b2_maxManifoldPoints];
    Vec2 normal;
    float nmxx, nmxy, nmyx, nmyy;   /* normalMass (b2Mat22): ex = (nmxx, nmxy), ey = (nmyx, nmyy) */
    float kxx, kxy, kyx, kyy;       /* K */
    int indexA, indexB;
    float invMassA, invMassB, invIA, invIB, friction, restitution;
    int pointCount, contact;
} VelocityConstraint;
typedef struct {
    Vec2 localPoints[b2_maxManifoldPoints], localNormal, localPoint;
    int indexA, indexB;
    float invMassA, invMassB;
    Vec2 localCenterA, localCenterB;
    float invIA, invIB;
    int type;
    float radiusA, radiusB;
    int pointCount;
} PositionConstraint;

typedef struct {
    int bodies[MWR_MAX_BODIES], bodyCount;
    int contacts[MWR_MAX_CONTACTS], contactCount;
    int joints[MWR_MAX_JOINTS], jointCount;
    Position positions[MWR_MAX_BODIES];
    Velocity velocities[MWR_MAX_BODIES];
    VelocityConstraint vc[MWR_MAX_CONTACTS];
    PositionConstraint pc[MWR_MAX_CONTACTS];
} Island;

static void solver_setup(World *w, Island *is, float dtRatio, int warmStarting) {   /* b2ContactSolver::b2ContactSolver */
    for (int i = 0; i < is->contactCount; ++i) {
        const Contact *contact = &w->contacts[is->contacts[i]];
        const Body *bodyA = &w->bodies[contact->bodyA], *bodyB = &w->bodies[contact->bodyB];
        const Manifold *manifold = &contact->manifold;
        VelocityConstraint *vc = &is->vc[i];
        vc->friction = contact->friction; vc->restitution = contact->restitution;
        vc->indexA = bodyA->islandIndex; vc->indexB = bodyB->islandIndex;
        vc->invMassA = bodyA->invMass; vc->invMassB = bodyB->invMass; vc->invIA = bodyA->invI; vc->invIB = bodyB->invI;
        vc->contact = i; vc->pointCount = manifold->pointCount;
        vc->kxx = vc->kxy = vc->kyx = vc->kyy = 0.0f; vc->nmxx = vc->nmxy = vc->nmyx = vc->nmyy = 0.0f;
        PositionConstraint *pc = &is->pc[i];
        pc->indexA = bodyA->islandIndex; pc->indexB = bodyB->islandIndex;
        pc->invMassA = bodyA->invMass; pc->invMassB = bodyB->invMass;
        pc->localCenterA = bodyA->sweep.localCenter; pc->localCenterB = bodyB->sweep.localCenter;
        pc->invIA = bodyA->invI; pc->invIB = bodyB->invI;
        pc->localNormal = manifold->localNormal; pc->localPoint = manifold->localPoint;
        pc->pointCount = manifold->pointCount;
        pc->radiusA = bodyA->shape.radius; pc->radiusB = bodyB->shape.radius;
        pc->type = manifold->type;
        for (int j = 0; j < manifold->pointCount; ++j) {
            const ManifoldPoint *cp = &manifold->points[j];
            VCPoint *vcp = &vc->points[j];
            if (warmStarting) { vcp->normalImpulse = dtRatio * cp->normalImpulse; vcp->tangentImpulse = dtRatio * cp->tangentImpulse; }
            else { vcp->normalImpulse = 0.0f; vcp->tangentImpulse = 0.0f; }
            vcp->rA = V(0, 0); vcp->rB = V(0, 0); vcp->normalMass = 0.0f; vcp->tangentMass = 0.0f; vcp->velocityBias = 0.0f;
            pc->localPoints[j] = cp->localPoint;
        }
    }
}
static void solver_init_velocity_constraints(World *w, Island *is) {   /* b2ContactSolver::InitializeVelocityConstraints */
    for (int i = 0; i < is->contactCount; ++i) {
        VelocityConstraint *vc = &is->vc[i];
        const PositionConstraint *pc = &is->pc[i];
        const Manifold *manifold = &w->contacts[is->contacts[vc->contact]].manifold;
        const int indexA = vc->indexA, indexB = vc->indexB;
        const float mA = vc->invMassA, mB = vc->invMassB, iA = vc->invIA, iB = vc->invIB;
        const Vec2 cA = is->positions[indexA].c, cB = is->positions[indexB].c;
        const float aA = is->positions[indexA].a, aB = is->positions[indexB].a;
        const Vec2 vA = is->velocities[indexA].v, vB = is->velocities[indexB].v;
        const float wA = is->velocities[indexA].w, wB = is->velocities[indexB].w;
        Xform xfA, xfB;
        xfA.q = rot_of(aA); xfB.q = rot_of(aB);
        xfA.p = vsub(cA, rmul(xfA.q, pc->localCenterA)); xfB.p = vsub(cB, rmul(xfB.q, pc->localCenterB));
        WorldManifold wm;
        world_manifold(&wm, manifold, xfA, pc->radiusA, xfB, pc->radiusB);
        vc->normal = wm.normal;
        for (int j = 0; j < vc->pointCount; ++j) {
            VCPoint *vcp = &vc->points[j];
            vcp->rA = vsub(wm.points[j], cA); vcp->rB = vsub(wm.points[j], cB);
            const float rnA = vcross(vcp->rA, vc->normal), rnB = vcross(vcp->rB, vc->normal);
            const float kNormal = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
            vcp->normalMass = kNormal > 0.0f ? 1.0f / kNormal : 0.0f;
            const Vec2 tangent = vcross_vs(vc->normal, 1.0f);
            const float rtA = vcross(vcp->rA, tangent), rtB = vcross(vcp->rB, tangent);
            const float kTangent = mA + mB + iA * rtA * rtA + iB * rtB * rtB;
            vcp->tangentMass = kTangent > 0.0f ? 1.0f / kTangent : 0.0f;
            vcp->velocityBias = 0.0f;   /* relative velocity bias for restitution */
            const float vRel = vdot(vc->normal, vsub(vsub(vadd(vB, vcross_sv(wB, vcp->rB)), vA), vcross_sv(wA, vcp->rA)));
            if (vRel < -b2_velocityThreshold) vcp->velocityBias = -vc->restitution * vRel;
        }
        if (vc->pointCount == 2) {      /* prepare the block solver */
            const VCPoint *vcp1 = &vc->points[0], *vcp2 = &vc->points[1];
            const float rn1A = vcross(vcp1->rA, vc->normal), rn1B = vcross(vcp1->rB, vc->normal);
            const float rn2A = vcross(vcp2->rA, vc->normal), rn2B = vcross(vcp2->rB, vc->normal);
            const float k11 = mA + mB + iA * rn1A * rn1A + iB * rn1B * rn1B;
            const float k22 = mA + mB + iA * rn2A * rn2A + iB * rn2B * rn2B;
            const float k12 = mA + mB + iA * rn1A * rn2A + iB * rn1B * rn2B;
            const float k_maxConditionNumber = 1000.0f;
            if (k11 * k11 < k_maxConditionNumber * (k11 * k22 - k12 * k12)) {   /* K is safe to invert */
                vc->kxx = k11; vc->kxy = k12; vc->kyx = k12; vc->kyy = k22;
                const float a = k11, b = k12, c = k12, d = k22;                  /* b2Mat22::GetInverse */
                float det = a * d - b * c;
                if (det != 0.0f) det = 1.0f / det;
                vc->nmxx = det * d; vc->nmyx = -det * b; vc->nmxy = -det * c; vc->nmyy = det * a;
            } else {
                vc->pointCount = 1;     /* the constraints are redundant, just use one */
            }
        }
    }
}
static void solver_warm_start(Island *is) {                /* b2ContactSolver::WarmStart */
    for (int i = 0; i < is->contactCount; ++i) {
        VelocityConstraint *vc = &is->vc[i];
        const int indexA = vc->indexA, indexB = vc->indexB;
        const float mA = vc->invMassA, iA = vc->invIA, mB = vc->invMassB, iB = vc->invIB;
        Vec2 vA = is->velocities[indexA].v, vB = is->velocities[indexB].v;
        float wA = is->velocities[indexA].w, wB = is->velocities[indexB].w;
        const Vec2 normal = vc->normal, tangent = vcross_vs(normal, 1.0f);
        for (int j = 0; j < vc->pointCount; ++j) {
            const VCPoint *vcp = &vc->points[j];
            const Vec2 P = vadd(vscale(vcp->normalImpulse, normal), vscale(vcp->tangentImpulse, tangent));
            wA -= iA * vcross(vcp->rA, P); vA = vsub(vA, vscale(mA, P));
            wB += iB * vcross(vcp->rB, P); vB = vadd(vB, vscale(mB, P));
        }
        is->velocities[indexA].v = vA; is->velocities[indexA].w = wA;
        is->velocities[indexB].v = vB; is->velocities[indexB].w = wB;
    }
}
static void solver_solve_velocity_constraints(Island *is) {   /* b2ContactSolver::SolveVelocityConstraints */
    for (int i = 0; i < is->contactCount; ++i) {
        VelocityConstraint *vc = &is->vc[i];
        const int indexA = vc->indexA, indexB = vc->indexB;
        const float mA = vc->invMassA, iA = vc->invIA, mB = vc->invMassB, iB = vc->invIB;
        const int pointCount = vc->pointCount;
        Vec2 vA = is->velocities[indexA].v, vB = is->velocities[indexB].v;
        float wA = is->velocities[indexA].w, wB = is->velocities[indexB].w;
        const Vec2 normal = vc->normal, tangent = vcross_vs(normal, 1.0f);
        const float friction = vc->friction;
        for (int j = 0; j < pointCount; ++j) {   /* tangent constraints first: non-penetration is more important than friction */
            VCPoint *vcp = &vc->points[j];
            const Vec2 dv = vsub(vsub(vadd(vB, vcross_sv(wB, vcp->rB)), vA), vcross_sv(wA, vcp->rA));
            const float vt = vdot(dv, tangent) - 0.0f /* tangentSpeed */;
            float lambda = vcp->tangentMass * (-vt);
            const float maxFriction = friction * vcp->normalImpulse;
            const float newImpulse = fclampf(vcp->tangentImpulse + lambda, -maxFriction, maxFriction);
            lambda = newImpulse - vcp->tangentImpulse;
            vcp->tangentImpulse = newImpulse;
            const Vec2 P = vscale(lambda, tangent);
            vA = vsub(vA, vscale(mA, P)); wA -= iA * vcross(vcp->rA, P);
            vB = vadd(vB, vscale(mB, P)); wB += iB * vcross(vcp->rB, P);
        }
        if (vc->pointCount == 1) {
            VCPoint *vcp = &vc->points[0];
            const Vec2 dv = vsub(vsub(vadd(vB, vcross_sv(wB, vcp->rB)), vA), vcross_sv(wA, vcp->rA));
            const float vn = vdot(dv, normal);
            float lambda = -vcp->normalMass * (vn - vcp->velocityBias);
            const float newImpulse = b2maxf(vcp->normalImpulse + lambda, 0.0f);
            lambda = newImpulse - vcp->normalImpulse;
            vcp->normalImpulse = newImpulse;
            const Vec2 P = vscale(lambda, normal);
            vA = vsub(vA, vscale(mA, P)); wA -= iA * vcross(vcp->rA, P);
            vB = vadd(vB, vscale(mB, P)); wB += iB * vcross(vcp->rB, P);
        } else {
            /* block solver: the 2-point LCP vn = A x + b', vn >= 0, x >= 0, vn_i x_i = 0, by enumeration of its four cases */
            VCPoint *cp1 = &vc->points[0], *cp2 = &vc->points[1];
            const Vec2 a = V(cp1->normalImpulse, cp2->normalImpulse);
            const Vec2 dv1 = vsub(vsub(vadd(vB, vcross_sv(wB, cp1->rB)), vA), vcross_sv(wA, cp1->rA));
            const Vec2 dv2 = vsub(vsub(vadd(vB, vcross_sv(wB, cp2->rB)), vA), vcross_sv(wA, cp2->rA));
            float vn1 = vdot(dv1, normal), vn2 = vdot(dv2, normal);
            Vec2 b = V(vn1 - cp1->velocityBias, vn2 - cp2->velocityBias);
            b = vsub(b, V(vc->kxx * a.x + vc->kyx * a.y, vc->kxy * a.x + vc->kyy * a.y));   /* b -= b2Mul(K, a) */
            for (;;) {
                Vec2 x = vneg(V(vc->nmxx * b.x + vc->nmyx * b.y, vc->nmxy * b.x + vc->nmyy * b.y));   /* case 1: vn = 0 */
                if (x.x >= 0.0f && x.y >= 0.0f) goto apply;
                x.x = -cp1->normalMass * b.x; x.y = 0.0f;                                               /* case 2: vn1 = 0 and x2 = 0 */
                vn1 = 0.0f; vn2 = vc->kxy * x.x + b.y;
                if (x.x >= 0.0f && vn2 >= 0.0f) goto apply;
                x.x = 0.0f; x.y = -cp2->normalMass * b.y;                                               /* case 3: vn2 = 0 and x1 = 0 */
                vn1 = vc->kyx * x.y + b.x; vn2 = 0.0f;
                if (x.y >= 0.0f && vn1 >= 0.0f) goto apply;
                x.x = 0.0f; x.y = 0.0f;                                                                 /* case 4: x1 = 0 and x2 = 0 */
                vn1 = b.x; vn2 = b.y;
                if (vn1 >= 0.0f && vn2 >= 0.0f) goto apply;
                break;   /* no solution, give up: hit when there is numerical error */
            apply: {
                    const Vec2 d = vsub(x, a);   /* incremental impulse */
                    const Vec2 P1 = vscale(d.x, normal), P2 = vscale(d.y, normal);
                    vA = vsub(vA, vscale(mA, vadd(P1, P2))); wA -= iA * (vcross(cp1->rA, P1) + vcross(cp2->rA, P2));
                    vB = vadd(vB, vscale(mB, vadd(P1, P2))); wB += iB * (vcross(cp1->rB, P1) + vcross(cp2->rB, P2));
                    cp1->normalImpulse = x.x; cp2->normalImpulse = x.y;
                }
                break;
            }
        }
        is->velocities[indexA].v = vA; is->velocities[indexA].w = wA;
        is->velocities[indexB].v = vB; is->velocities[indexB].w = wB;
    }
}
static void solver_store_impulses(World *w, Island *is) {   /* b2ContactSolver::StoreImpulses */
    for (int i = 0; i < is->contactCount; ++i) {
        const VelocityConstraint *vc = &is->vc[i];
        Manifold *manifold = &w->contacts[is->contacts[vc->contact]].manifold;
        for (int j = 0; j < vc->pointCount; ++j) {
            manifold->points[j].normalImpulse = vc->points[j].normalImpulse;
            manifold->points[j].tangentImpulse = vc->points[j].tangentImpulse;
        }
    }
}
/* b2PositionSolverManifold::Initialize */
static void position_manifold(const PositionConstraint *pc, Xform xfA, Xform xfB, int index, Vec2 *normal, Vec2 *point, float *separation) {
    if (pc->type == MF_FACE_A) {
        *normal = rmul(xfA.q, pc->localNormal);
        const Vec2 planePoint = xmul(xfA, pc->localPoint), clipPoint = xmul(xfB, pc->localPoints[index]);
        *separation = vdot(vsub(clipPoint, planePoint), *normal) - pc->radiusA - pc->radiusB;
        *point = clipPoint;
    } else {
        *normal = rmul(xfB.q, pc->localNormal);
        const Vec2 planePoint = xmul(xfB, pc->localPoint), clipPoint = xmul(xfA, pc->localPoints[index]);
        *separation = vdot(vsub(clipPoint, planePoint), *normal) - pc->radiusA - pc->radiusB;
        *point = clipPoint;
        *normal = vneg(*normal);   /* ensure the normal points from A to B */
    }
}
/* b2ContactSolver::SolvePositionConstraints (toiIndexA < 0) and SolveTOIPositionConstraints (only the two TOI bodies move) */
static int solver_solve_position_constraints(Island *is, int toiIndexA, int toiIndexB) {
    const int toi = toiIndexA >= 0;
    float minSeparation = 0.0f;
    for (int i = 0; i < is->contactCount; ++i) {
        const PositionConstraint *pc = &is->pc[i];
        const int indexA = pc->indexA, indexB = pc->indexB;
        const Vec2 localCenterA = pc->localCenterA, localCenterB = pc->localCenterB;
        float mA = pc->invMassA, iA = pc->invIA, mB = pc->invMassB, iB = pc->invIB;
        if (toi) {
            if (!(indexA == toiIndexA || indexA == toiIndexB)) { mA = 0.0f; iA = 0.0f; }
            if (!(indexB == toiIndexA || indexB == toiIndexB)) { mB = 0.0f; iB = 0.0f; }
        }
        Vec2 cA = is->positions[indexA].c, cB = is->positions[indexB].c;
        float aA = is->positions[indexA].a, aB = is->positions[indexB].a;
        for (int j = 0; j < pc->pointCount; ++j) {   /* solve normal constraints */
            Xform xfA, xfB;
            xfA.q = rot_of(aA); xfB.q = rot_of(aB);
            xfA.p = vsub(cA, rmul(xfA.q, localCenterA)); xfB.p = vsub(cB, rmul(xfB.q, localCenterB));
            Vec2 normal, point;
            float separation;
            position_manifold(pc, xfA, xfB, j, &normal, &point, &separation);
            const Vec2 rA = vsub(point, cA), rB = vsub(point, cB);
            minSeparation = b2minf(minSeparation, separation);   /* track max constraint error */
            const float C = fclampf((toi ? b2_toiBaugarte : b2_baumgarte) * (separation + b2_linearSlop), -b2_maxLinearCorrection, 0.0f);
            const float rnA = vcross(rA, normal), rnB = vcross(rB, normal);
            const float K = mA + mB + iA * rnA * rnA + iB * rnB * rnB;
            const float impulse = K > 0.0f ? -C / K : 0.0f;
            const Vec2 P = vscale(impulse, normal);
            cA = vsub(cA, vscale(mA, P)); aA -= iA * vcross(rA, P);
            cB = vadd(cB, vscale(mB, P)); aB += iB * vcross(rB, P);
        }
        is->positions[indexA].c = cA; is->positions[indexA].a = aA;
        is->positions[indexB].c = cB; is->positions[indexB].a = aB;
    }
    /* we can't expect minSeparation >= -b2_linearSlop because we don't push the separation above -b2_linearSlop */
    return minSeparation >= (toi ? -1.5f : -3.0f) * b2_linearSlop;
}

/* ------------------------------------------------------------------------------------------------ b2RevoluteJoint */
static void mat33_solve33(const float *m, float bx, float by, float bz, float *x, float *y, float *z) {
    const float exx = m[0], exy = m[1], exz = m[2], eyx = m[3], eyy = m[4], eyz = m[5], ezx = m[6], ezy = m[7], ezz = m[8];
    /* det = b2Dot(ex, b2Cross(ey, ez)) */
    const float cx = eyy * ezz - eyz * ezy, cy = eyz * ezx - eyx * ezz, cz = eyx * ezy - eyy * ezx;
    float det = exx * cx + exy * cy + exz * cz;
    if (det != 0.0f) det = 1.0f / det;
    *x = det * (bx * cx + by * cy + bz * cz);                               /* b2Dot(b, b2Cross(ey, ez)) */
    const float dx = by * ezz - bz * ezy, dy = bz * ezx - bx * ezz, dz = bx * ezy - by * ezx;   /* b2Cross(b, ez) */
    *y = det * (exx * dx + exy * dy + exz * dz);
    const float fx = eyy * bz - eyz * by, fy = eyz * bx - eyx * bz, fz = eyx * by - eyy * bx;   /* b2Cross(ey, b) */
    *z = det * (exx * fx + exy * fy + exz * fz);
}
static void mat33_solve22(const float *m, float bx, float by, float *x, float *y) {
    const float a11 = m[0], a12 = m[3], a21 = m[1], a22 = m[4];
    float det = a11 * a22 - a12 * a21;
    if (det != 0.0f) det = 1.0f / det;
    *x = det * (a22 * bx - a12 * by);
    *y = det * (a11 * by - a21 * bx);
}
static void joint_init_velocity_constraints(World *w, Island *is, RevoluteJoint *j, float dtRatio, int warmStarting) {
    const Body *bA = &w->bodies[j->bodyA], *bB = &w->bodies[j->bodyB];
    j->indexA = bA->islandIndex; j->indexB = bB->islandIndex;
    j->localCenterA = bA->sweep.localCenter; j->localCenterB = bB->sweep.localCenter;
    j->invMassA = bA->invMass; j->invMassB = bB->invMass; j->invIA = bA->invI; j->invIB = bB->invI;
    const float aA = is->positions[j->indexA].a, aB = is->positions[j->indexB].a;
    Vec2 vA = is->velocities[j->indexA].v, vB = is->velocities[j->indexB].v;
    float wA = is->velocities[j->indexA].w, wB = is->velocities[j->indexB].w;
    const Rot qA = rot_of(aA), qB = rot_of(aB);
    j->rA = rmul(qA, vsub(j->localAnchorA, j->localCenterA));
    j->rB = rmul(qB, vsub(j->localAnchorB, j->localCenterB));
    const float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
    const int fixedRotation = (iA + iB == 0.0f);
    float *m = j->mass;
    m[0] = mA + mB + j->rA.y * j->rA.y * iA + j->rB.y * j->rB.y * iB;   /* ex.x */
    m[3] = -j->rA.y * j->rA.x * iA - j->rB.y * j->rB.x * iB;            /* ey.x */
    m[6] = -j->rA.y * iA - j->rB.y * iB;                                /* ez.x */
    m[1] = m[3];                                                        /* ex.y */
    m[4] = mA + mB + j->rA.x * j->rA.x * iA + j->rB.x * j->rB.x * iB;   /* ey.y */
    m[7] = j->rA.x * iA + j->rB.x * iB;                                 /* ez.y */
    m[2] = m[6]; m[5] = m[7];                                           /* ex.z, ey.z */
    m[8] = iA + iB;                                                     /* ez.z */
    j->motorMass = iA + iB;
    if (j->motorMass > 0.0f) j->motorMass = 1.0f / j->motorMass;
    if (!j->enableMotor || fixedRotation) j->motorImpulse = 0.0f;
    if (j->enableLimit && !fixedRotation) {
        const float jointAngle = aB - aA - j->referenceAngle;
        if (fabsf(j->upperAngle - j->lowerAngle) < 2.0f * b2_angularSlop) j->limitState = LIMIT_EQUAL;
        else if (jointAngle <= j->lowerAngle) { if (j->limitState != LIMIT_AT_LOWER) j->impulse[2] = 0.0f; j->limitState = LIMIT_AT_LOWER; }
        else if (jointAngle >= j->upperAngle) { if (j->limitState != LIMIT_AT_UPPER) j->impulse[2] = 0.0f; j->limitState = LIMIT_AT_UPPER; }
        else { j->limitState = LIMIT_INACTIVE; j->impulse[2] = 0.0f; }
    } else j->limitState = LIMIT_INACTIVE;
    if (warmStarting) {
        j->impulse[0] *= dtRatio; j->impulse[1] *= dtRatio; j->impulse[2] *= dtRatio;   /* scale impulses to support a variable time step */
        j->motorImpulse *= dtRatio;
        const Vec2 P = V(j->impulse[0], j->impulse[1]);
        vA = vsub(vA, vscale(mA, P)); wA -= iA * (vcross(j->rA, P) + j->motorImpulse + j->impulse[2]);
        vB = vadd(vB, vscale(mB, P)); wB += iB * (vcross(j->rB, P) + j->motorImpulse + j->impulse[2]);
    } else { j->impulse[0] = j->impulse[1] = j->impulse[2] = 0.0f; j->motorImpulse = 0.0f; }
    is->velocities[j->indexA].v = vA; is->velocities[j->indexA].w = wA;
    is->velocities[j->indexB].v = vB; is->velocities[j->indexB].w = wB;
}
static void joint_solve_velocity_constraints(Island *is, RevoluteJoint *j, float dt) {
    Vec2 vA = is->velocities[j->indexA].v, vB = is->velocities[j->indexB].v;
    float wA = is->velocities[j->indexA].w, wB = is->velocities[j->indexB].w;
    const float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
    const int fixedRotation = (iA + iB == 0.0f);
    if (j->enableMotor && j->limitState != LIMIT_EQUAL && !fixedRotation) {   /* motor constraint */
        const float Cdot = wB - wA - j->motorSpeed;
        float impulse = -j->motorMass * Cdot;
        const float oldImpulse = j->motorImpulse, maxImpulse = dt * j->maxMotorTorque;
        j->motorImpulse = fclampf(oldImpulse + impulse, -maxImpulse, maxImpulse);
        impulse = j->motorImpulse - oldImpulse;
        wA -= iA * impulse; wB += iB * impulse;
    }
    if (j->enableLimit && j->limitState != LIMIT_INACTIVE && !fixedRotation) {   /* limit constraint: 3x3 */
        const Vec2 Cdot1 = vsub(vsub(vadd(vB, vcross_sv(wB, j->rB)), vA), vcross_sv(wA, j->rA));
        const float Cdot2 = wB - wA;
        float ix, iy, iz;
        mat33_solve33(j->mass, Cdot1.x, Cdot1.y, Cdot2, &ix, &iy, &iz);
        ix = -ix; iy = -iy; iz = -iz;
        if (j->limitState == LIMIT_EQUAL) { j->impulse[0] += ix; j->impulse[1] += iy; j->impulse[2] += iz; }
        else {
            const float newImpulse = j->impulse[2] + iz;
            const int reduce = j->limitState == LIMIT_AT_LOWER ? newImpulse < 0.0f : newImpulse > 0.0f;
            if (reduce) {
                const float rhsx = -Cdot1.x + j->impulse[2] * j->mass[6], rhsy = -Cdot1.y + j->impulse[2] * j->mass[7];
                float rx, ry;
                mat33_solve22(j->mass, rhsx, rhsy, &rx, &ry);
                ix = rx; iy = ry; iz = -j->impulse[2];
                j->impulse[0] += rx; j->impulse[1] += ry; j->impulse[2] = 0.0f;
            } else { j->impulse[0] += ix; j->impulse[1] += iy; j->impulse[2] += iz; }
        }
        const Vec2 P = V(ix, iy);
        vA = vsub(vA, vscale(mA, P)); wA -= iA * (vcross(j->rA, P) + iz);
        vB = vadd(vB, vscale(mB, P)); wB += iB * (vcross(j->rB, P) + iz);
    } else {                                                                    /* point-to-point constraint */
        const Vec2 Cdot = vsub(vsub(vadd(vB, vcross_sv(wB, j->rB)), vA), vcross_sv(wA, j->rA));
        float ix, iy;
        mat33_solve22(j->mass, -Cdot.x, -Cdot.y, &ix, &iy);
        j->impulse[0] += ix; j->impulse[1] += iy;
        const Vec2 impulse = V(ix, iy);
        vA = vsub(vA, vscale(mA, impulse)); wA -= iA * vcross(j->rA, impulse);
        vB = vadd(vB, vscale(mB, impulse)); wB += iB * vcross(j->rB, impulse);
    }
    is->velocities[j->indexA].v = vA; is->velocities[j->indexA].w = wA;
    is->velocities[j->indexB].v = vB; is->velocities[j->indexB].w = wB;
}
static int joint_solve_position_constraints(Island *is, RevoluteJoint *j) {
    Vec2 cA = is->positions[j->indexA].c, cB = is->positions[j->indexB].c;
    float aA = is->positions[j->indexA].a, aB = is->positions[j->indexB].a;
    float angularError = 0.0f, positionError = 0.0f;
    const int fixedRotation = (j->invIA + j->invIB == 0.0f);
    if (j->enableLimit && j->limitState != LIMIT_INACTIVE && !fixedRotation) {   /* angular limit constraint */
        const float angle = aB - aA - j->referenceAngle;
        float limitImpulse = 0.0f;
        if (j->limitState == LIMIT_EQUAL) {
            const float C = fclampf(angle - j->lowerAngle, -b2_maxAngularCorrection, b2_maxAngularCorrection);   /* prevent large angular corrections */
            limitImpulse = -j->motorMass * C;
            angularError = fabsf(C);
        } else if (j->limitState == LIMIT_AT_LOWER) {
            float C = angle - j->lowerAngle;
            angularError = -C;
            C = fclampf(C + b2_angularSlop, -b2_maxAngularCorrection, 0.0f);   /* prevent large angular corrections and allow some slop */
            limitImpulse = -j->motorMass * C;
        } else {
            float C = angle - j->upperAngle;
            angularError = C;
            C = fclampf(C - b2_angularSlop, 0.0f, b2_maxAngularCorrection);
            limitImpulse = -j->motorMass * C;
        }
        aA -= j->invIA * limitImpulse; aB += j->invIB * limitImpulse;
    }
    {   /* point-to-point constraint */
        const Rot qA = rot_of(aA), qB = rot_of(aB);
        const Vec2 rA = rmul(qA, vsub(j->localAnchorA, j->localCenterA)), rB = rmul(qB, vsub(j->localAnchorB, j->localCenterB));
        const Vec2 C = vsub(vsub(vadd(cB, rB), cA), rA);
        positionError = vlen(C);
        const float mA = j->invMassA, mB = j->invMassB, iA = j->invIA, iB = j->invIB;
        const float Kexx = mA + mB + iA * rA.y * rA.y + iB * rB.y * rB.y;
        const float Kexy = -iA * rA.x * rA.y - iB * rB.x * rB.y;
        const float Keyx = Kexy;
        const float Keyy = mA + mB + iA * rA.x * rA.x + iB * rB.x * rB.x;
        /* impulse = -K.Solve(C), b2Mat22::Solve */
        const float a11 = Kexx, a12 = Keyx, a21 = Kexy, a22 = Keyy;
        float det = a11 * a22 - a12 * a21;
        if (det != 0.0f) det = 1.0f / det;
        const Vec2 impulse = V(-(det * (a22 * C.x - a12 * C.y)), -(det * (a11 * C.y - a21 * C.x)));
        cA = vsub(cA, vscale(mA, impulse)); aA -= iA * vcross(rA, impulse);
        cB = vadd(cB, vscale(mB, impulse)); aB += iB * vcross(rB, impulse);
    }
    is->positions[j->indexA].c = cA; is->positions[j->indexA].a = aA;
    is->positions[j->indexB].c = cB; is->positions[j->indexB].a = aB;
    return positionError <= b2_linearSlop && angularError <= b2_angularSlop;
}

/* ------------------------------------------------------------------------------------------------ b2Island */
static void island_add_body(World *w, Island *is, int b) { w->bodies[b].islandIndex = is->bodyCount; is->bodies[is->bodyCount++] = b; }

static void island_solve(World *w, Island *is, float dt, float dtRatio, int velocityIterations, int positionIterations) {   /* b2Island::Solve */
    const float h = dt;
    for (int i = 0; i < is->bodyCount; ++i) {   /* integrate velocities */
        Body *b = &w->bodies[is->bodies[i]];
        const Vec2 c = b->sweep.c;
        const float a = b->sweep.a;
        Vec2 v = b->linearVelocity;
        float wv = b->angularVelocity;
        b->sweep.c0 = b->sweep.c; b->sweep.a0 = b->sweep.a;   /* store positions for continuous collision */
        if (b->type == BODY_DYNAMIC) {
            v = vadd(v, vscale(h, vadd(vscale(1.0f /* gravityScale */, w->gravity), vscale(b->invMass, b->force))));
            wv += h * b->invI * b->torque;
            /* linear / angular damping are 0: v *= 1 / (1 + h * 0) */
            v = vscale(1.0f / (1.0f + h * 0.0f), v);
            wv *= 1.0f / (1.0f + h * 0.0f);
        }
        is->positions[i].c = c; is->positions[i].a = a;
        is->velocities[i].v = v; is->velocities[i].w = wv;
    }
    solver_setup(w, is, dtRatio, w->warmStarting);
    solver_init_velocity_constraints(w, is);
    if (w->warmStarting) solver_warm_start(is);
    for (int i = 0; i < is->jointCount; ++i) joint_init_velocity_constraints(w, is, &w->joints[is->joints[i]], dtRatio, w->warmStarting);
    for (int it = 0; it < velocityIterations; ++it) {   /* solve velocity constraints */
        for (int i = 0; i < is->jointCount; ++i) joint_solve_velocity_constraints(is, &w->joints[is->joints[i]], dt);
        solver_solve_velocity_constraints(is);
    }
    solver_store_impulses(w, is);                       /* for warm starting */
    for (int i = 0; i < is->bodyCount; ++i) {           /* integrate positions */
        Vec2 c = is->positions[i].c;
        float a = is->positions[i].a;
        Vec2 v = is->velocities[i].v;
        float wv = is->velocities[i].w;
        const Vec2 translation = vscale(h, v);
        if (vdot(translation, translation) > b2_maxTranslationSquared) v = vscale(b2_maxTranslation / vlen(translation), v);
        const float rotation = h * wv;
        if (rotation * rotation > b2_maxRotationSquared) wv *= b2_maxRotation / fabsf(rotation);
        c = vadd(c, vscale(h, v));
        a += h * wv;
        is->positions[i].c = c; is->positions[i].a = a;
        is->velocities[i].v = v; is->velocities[i].w = wv;
    }
    int positionSolved = 0;
    for (int it = 0; it < positionIterations; ++it) {   /* solve position constraints */
        const int contactsOkay = solver_solve_position_constraints(is, -1, -1);
        int jointsOkay = 1;
        for (int i = 0; i < is->jointCount; ++i) {
            const int jointOkay = joint_solve_position_constraints(is, &w->joints[is->joints[i]]);
            jointsOkay = jointsOkay && jointOkay;
        }
        if (contactsOkay && jointsOkay) { positionSolved = 1; break; }   /* exit early if the position errors are small */
    }
    for (int i = 0; i < is->bodyCount; ++i) {           /* copy state buffers back to the bodies */
        Body *b = &w->bodies[is->bodies[i]];
        b->sweep.c = is->positions[i].c; b->sweep.a = is->positions[i].a;
        b->linearVelocity = is->velocities[i].v; b->angularVelocity = is->velocities[i].w;
        body_sync_transform(b);
    }
    if (w->allowSleep) {
        float minSleepTime = b2_maxFloat;
        const float linTolSqr = b2_linearSleepTolerance * b2_linearSleepTolerance;
        const float angTolSqr = b2_angularSleepTolerance * b2_angularSleepTolerance;
        for (int i = 0; i < is->bodyCount; ++i) {
            Body *b = &w->bodies[is->bodies[i]];
            if (b->type == BODY_STATIC) continue;
            if (b->angularVelocity * b->angularVelocity > angTolSqr || vdot(b->linearVelocity, b->linearVelocity) > linTolSqr) {
                b->sleepTime = 0.0f; minSleepTime = 0.0f;
            } else {
                b->sleepTime += h;
                minSleepTime = b2minf(minSleepTime, b->sleepTime);
            }
        }
        if (minSleepTime >= b2_timeToSleep && positionSolved)
            for (int i = 0; i < is->bodyCount; ++i) body_set_awake(&w->bodies[is->bodies[i]], 0);
    }
}

static void island_solve_toi(World *w, Island *is, float subDt, int velocityIterations, int toiIndexA, int toiIndexB) {   /* b2Island::SolveTOI */
    for (int i = 0; i < is->bodyCount; ++i) {   /* initialize the body state */
        const Body *b = &w->bodies[is->bodies[i]];
        is->positions[i].c = b->sweep.c; is->positions[i].a = b->sweep.a;
        is->velocities[i].v = b->linearVelocity; is->velocities[i].w = b->angularVelocity;
    }
    solver_setup(w, is, 1.0f, 0);               /* subStep.warmStarting = false */
    for (int it = 0; it < 20; ++it)             /* subStep.positionIterations = 20 */
        if (solver_solve_position_constraints(is, toiIndexA, toiIndexB)) break;
    /* leap of faith to new safe state */
    w->bodies[is->bodies[toiIndexA]].sweep.c0 = is->positions[toiIndexA].c; w->bodies[is->bodies[toiIndexA]].sweep.a0 = is->positions[toiIndexA].a;
    w->bodies[is->bodies[toiIndexB]].sweep.c0 = is->positions[toiIndexB].c; w->bodies[is->bodies[toiIndexB]].sweep.a0 = is->positions[toiIndexB].a;
    /* no warm starting is needed for TOI events because warm starting impulses were applied in the discrete solver */
    solver_init_velocity_constraints(w, is);
    for (int it = 0; it < velocityIterations; ++it) solver_solve_velocity_constraints(is);
    /* don't store the TOI contact forces for warm starting because they can be quite large */
    const float h = subDt;
    for (int i = 0; i < is->bodyCount; ++i) {   /* integrate positions */
        Vec2 c = is->positions[i].c;
        float a = is->positions[i].a;
        Vec2 v = is->velocities[i].v;
        float wv = is->velocities[i].w;
        const Vec2 translation = vscale(h, v);
        if (vdot(translation, translation) > b2_maxTranslationSquared) v = vscale(b2_maxTranslation / vlen(translation), v);
        const float rotation = h * wv;
        if (rotation * rotation > b2_maxRotationSquared) wv *= b2_maxRotation / fabsf(rotation);
        c = vadd(c, vscale(h, v));
        a += h * wv;
        is->positions[i].c = c; is->positions[i].a = a;
        is->velocities[i].v = v; is->velocities[i].w = wv;
        Body *b = &w->bodies[is->bodies[i]];    /* sync bodies */
        b->sweep.c = c; b->sweep.a = a; b->linearVelocity = v; b->angularVelocity = wv;
        body_sync_transform(b);
    }
}

/* ------------------------------------------------------------------------------------------------ b2World */
static void world_solve(World *w, float dt, float dtRatio, int velocityIterations, int positionIterations) {   /* b2World::Solve */
    static __thread Island island;
    for (int b = 0; b < w->bodyCount; ++b) w->bodies[b].islandFlag = 0;
    for (int c = w->contactList; c >= 0; c = w->contacts[c].next) w->contacts[c].islandFlag = 0;
    for (int j = 0; j < w->jointCount; ++j) w->joints[j].islandFlag = 0;
    int stack[MWR_MAX_BODIES];
    for (int seed = w->bodyList; seed >= 0; seed = w->bodies[seed].next) {   /* build and simulate all awake islands */
        Body *sb = &w->bodies[seed];
        if (sb->islandFlag) continue;
        if (!sb->awake) continue;
        if (sb->type == BODY_STATIC) continue;     /* the seed can be dynamic or kinematic */
        island.bodyCount = island.contactCount = island.jointCount = 0;
        int stackCount = 0;
        stack[stackCount++] = seed;
        sb->islandFlag = 1;
        while (stackCount > 0) {                   /* depth first search on the constraint graph */
            const int bi = stack[--stackCount];
            Body *b = &w->bodies[bi];
            island_add_body(w, &island, bi);
            body_set_awake(b, 1);                  /* make sure the body is awake */
            if (b->type == BODY_STATIC) continue;  /* don't propagate islands across static bodies */
            for (int e = b->contactList; e >= 0; e = w->contacts[e >> 1].edgeNext[e & 1]) {   /* all contacts on the body */
                Contact *contact = &w->contacts[e >> 1];
                if (contact->islandFlag) continue;                       /* already added to an island */
                if (!contact->enabled || !contact->touching) continue;  /* is this contact solid and touching? */
                island.contacts[island.contactCount++] = e >> 1;
                contact->islandFlag = 1;
                const int other = (e & 1) ? contact->bodyA : contact->bodyB;
                if (w->bodies[other].islandFlag) continue;               /* was the other body already added to this island? */
                stack[stackCount++] = other;
                w->bodies[other].islandFlag = 1;
            }
            for (int e = b->jointList; e >= 0; e = w->joints[e >> 1].edgeNext[e & 1]) {       /* all joints connected to the body */
                RevoluteJoint *joint = &w->joints[e >> 1];
                if (joint->islandFlag) continue;
                const int other = (e & 1) ? joint->bodyA : joint->bodyB;
                island.joints[island.jointCount++] = e >> 1;
                joint->islandFlag = 1;
                if (w->bodies[other].islandFlag) continue;
                stack[stackCount++] = other;
                w->bodies[other].islandFlag = 1;
            }
        }
        island_solve(w, &island, dt, dtRatio, velocityIterations, positionIterations);
        for (int i = 0; i < island.bodyCount; ++i) {   /* post solve cleanup: allow static bodies to participate in other islands */
            Body *b = &w->bodies[island.bodies[i]];
            if (b->type == BODY_STATIC) b->islandFlag = 0;
        }
    }
    for (int b = w->bodyList; b >= 0; b = w->bodies[b].next) {   /* synchronize fixtures */
        Body *body = &w->bodies[b];
        if (!body->islandFlag) continue;           /* if a body was not in an island then it did not move */
        if (body->type == BODY_STATIC) continue;
        body_sync_fixtures(w, body);               /* update fixtures (for broad-phase) */
    }
    contact_manager_find_new_contacts(w);          /* look for new contacts */
}

static void world_solve_toi(World *w, float dt, int velocityIterations) {   /* b2World::SolveTOI */
    static __thread Island island;
    if (w->stepComplete) {
        for (int b = 0; b < w->bodyCount; ++b) { w->bodies[b].islandFlag = 0; w->bodies[b].sweep.alpha0 = 0.0f; }
        for (int c = w->contactList; c >= 0; c = w->contacts[c].next) {   /* invalidate TOI */
            Contact *k = &w->contacts[c];
            k->toiFlag = 0; k->islandFlag = 0; k->toiCount = 0; k->toi = 1.0f;
        }
    }
    for (;;) {   /* find TOI events and solve them */
        int minContact = -1;
        float minAlpha = 1.0f;
        for (int ci = w->contactList; ci >= 0; ci = w->contacts[ci].next) {
            Contact *c = &w->contacts[ci];
            if (!c->enabled) continue;                        /* is this contact disabled? */
            if (c->toiCount > b2_maxSubSteps) continue;       /* prevent excessive sub-stepping */
            float alpha = 1.0f;
            if (c->toiFlag) {
                alpha = c->toi;                               /* this contact has a valid cached TOI */
            } else {
                Body *bA = &w->bodies[c->bodyA], *bB = &w->bodies[c->bodyB];
                const int activeA = bA->awake && bA->type != BODY_STATIC, activeB = bB->awake && bB->type != BODY_STATIC;
                if (!activeA && !activeB) continue;           /* is at least one body active (awake and dynamic or kinematic)? */
                const int collideA = /* bullet */ 0 || bA->type != BODY_DYNAMIC, collideB = 0 || bB->type != BODY_DYNAMIC;
                if (!collideA && !collideB) continue;         /* are these two non-bullet dynamic bodies? */
                float alpha0 = bA->sweep.alpha0;              /* put the sweeps onto the same time interval */
                if (bA->sweep.alpha0 < bB->sweep.alpha0) { alpha0 = bB->sweep.alpha0; sweep_advance(&bA->sweep, alpha0); }
                else if (bB->sweep.alpha0 < bA->sweep.alpha0) { alpha0 = bA->sweep.alpha0; sweep_advance(&bB->sweep, alpha0); }
                const DProxy pA = dproxy_of(&bA->shape), pB = dproxy_of(&bB->shape);
                float beta;                                   /* the fraction of the remaining portion of the step */
                const int state = time_of_impact(&beta, &pA, &bA->sweep, &pB, &bB->sweep, 1.0f);
                if (state == TOI_TOUCHING) alpha = b2minf(alpha0 + (1.0f - alpha0) * beta, 1.0f);
                else alpha = 1.0f;
                c->toi = alpha;
                c->toiFlag = 1;
            }
            if (alpha < minAlpha) { minContact = ci; minAlpha = alpha; }   /* this is the minimum TOI found so far */
        }
        if (minContact < 0 || 1.0f - 10.0f * b2_epsilon < minAlpha) { w->stepComplete = 1; break; }   /* no more TOI events: done */
        Contact *mc = &w->contacts[minContact];
        const int iA = mc->bodyA, iB = mc->bodyB;
        Body *bA = &w->bodies[iA], *bB = &w->bodies[iB];
        const Sweep backup1 = bA->sweep, backup2 = bB->sweep;
        body_advance(bA, minAlpha); body_advance(bB, minAlpha);   /* advance the bodies to the TOI */
        contact_update(w, minContact);                            /* the TOI contact likely has some new contact points */
        mc->toiFlag = 0;
        ++mc->toiCount;
        ++w->stat_toi_events;
        if (!mc->enabled || !mc->touching) {                      /* is the contact solid? */
            mc->enabled = 0;                                      /* restore the sweeps */
            bA->sweep = backup1; bB->sweep = backup2;
            body_sync_transform(bA); body_sync_transform(bB);
            continue;
        }
        body_set_awake(bA, 1); body_set_awake(bB, 1);
        island.bodyCount = island.contactCount = island.jointCount = 0;   /* build the island */
        island_add_body(w, &island, iA); island_add_body(w, &island, iB);
        island.contacts[island.contactCount++] = minContact;
        bA->islandFlag = 1; bB->islandFlag = 1; mc->islandFlag = 1;
        const int pair[2] = {iA, iB};
        for (int i = 0; i < 2; ++i) {                             /* get contacts on bodyA and bodyB */
            const int bi = pair[i];
            Body *body = &w->bodies[bi];
            if (body->type != BODY_DYNAMIC) continue;
            for (int e = body->contactList; e >= 0; e = w->contacts[e >> 1].edgeNext[e & 1]) {
                if (island.bodyCount == 2 * b2_maxTOIContacts) break;
                if (island.contactCount == b2_maxTOIContacts) break;
                const int ci = e >> 1;
                Contact *contact = &w->contacts[ci];
                if (contact->islandFlag) continue;                /* already added to the island */
                const int oi = (e & 1) ? contact->bodyA : contact->bodyB;
                Body *other = &w->bodies[oi];
                if (other->type == BODY_DYNAMIC /* && neither is a bullet */) continue;   /* only add static, kinematic, or bullet bodies */
                const Sweep backup = other->sweep;                /* tentatively advance the body to the TOI */
                if (!other->islandFlag) body_advance(other, minAlpha);
                contact_update(w, ci);                            /* update the contact points */
                if (!contact->enabled) { other->sweep = backup; body_sync_transform(other); continue; }    /* disabled by the user */
                if (!contact->touching) { other->sweep = backup; body_sync_transform(other); continue; }   /* no contact points */
                contact->islandFlag = 1;                          /* add the contact to the island */
                island.contacts[island.contactCount++] = ci;
                if (other->islandFlag) continue;                  /* the other body is already in the island */
                other->islandFlag = 1;
                if (other->type != BODY_STATIC) body_set_awake(other, 1);
                island_add_body(w, &island, oi);
            }
        }
        const float subDt = (1.0f - minAlpha) * dt;
        island_solve_toi(w, &island, subDt, velocityIterations, bA->islandIndex, bB->islandIndex);
        for (int i = 0; i < island.bodyCount; ++i) {              /* reset island flags and synchronize broad-phase proxies */
            Body *body = &w->bodies[island.bodies[i]];
            body->islandFlag = 0;
            if (body->type != BODY_DYNAMIC) continue;
            body_sync_fixtures(w, body);
            for (int e = body->contactList; e >= 0; e = w->contacts[e >> 1].edgeNext[e & 1]) {   /* invalidate all contact TOIs on this displaced body */
                w->contacts[e >> 1].toiFlag = 0; w->contacts[e >> 1].islandFlag = 0;
            }
        }
        contact_manager_find_new_contacts(w);   /* commit fixture proxy movements to the broad-phase so that new contacts are created */
    }
}

static void world_step(World *w, float dt, int velocityIterations, int positionIterations) {   /* b2World::Step */
    if (w->newFixture) { contact_manager_find_new_contacts(w); w->newFixture = 0; }   /* new fixtures were added: find the new contacts */
    const float inv_dt = dt > 0.0f ? 1.0f / dt : 0.0f;
    const float dtRatio = w->inv_dt0 * dt;
    contact_manager_collide(w);                                    /* update contacts: this is where some contacts are destroyed */
    if (w->stepComplete && dt > 0.0f) world_solve(w, dt, dtRatio, velocityIterations, positionIterations);   /* integrate, solve velocity constraints, integrate positions */
    if (w->continuousPhysics && dt > 0.0f) world_solve_toi(w, dt, velocityIterations);   /* handle TOI events */
    if (dt > 0.0f) w->inv_dt0 = inv_dt;
    for (int b = 0; b < w->bodyCount; ++b) { w->bodies[b].force = V(0, 0); w->bodies[b].torque = 0.0f; }   /* ClearForces (e_clearForces default) */
}

/* b2World::RayCast over the edge fixtures, closest hit (D2); b2EdgeShape::RayCast per fixture.  Returns the fraction, 1.0 without a hit. */
static float world_raycast_closest(const World *w_, Vec2 p1w, Vec2 p2w, uint16_t categoryMask) {
    World *w = (World *)w_;   /* (the counters) */
    int hits = 0;
    float best = 1.0f;   /* LidarCallback.fraction starts at 1.0 (:210); input.maxFraction = 1 */
    for (int bi = 0; bi < w->bodyCount; ++bi) {
        const Body *b = &w->bodies[bi];
        if (b->shape.type != SHAPE_EDGE || (b->categoryBits & categoryMask) == 0) continue;
        const Vec2 p1 = rmulT(b->xf.q, vsub(p1w, b->xf.p)), p2 = rmulT(b->xf.q, vsub(p2w, b->xf.p));   /* the ray in the edge's frame */
        const Vec2 d = vsub(p2, p1);
        const Vec2 v1 = b->shape.v[0], v2 = b->shape.v[1];
        const Vec2 e = vsub(v2, v1);
        Vec2 normal = V(e.y, -e.x);
        vnormalize(&normal);
        const float numerator = vdot(normal, vsub(v1, p1)), denominator = vdot(normal, d);
        if (denominator == 0.0f) continue;
        const float t = numerator / denominator;
        if (t < 0.0f || 1.0f < t) continue;
        const Vec2 q = vadd(p1, vscale(t, d));
        const Vec2 r = vsub(v2, v1);
        const float rr = vdot(r, r);
        if (rr == 0.0f) continue;
        const float s = vdot(vsub(q, v1), r) / rr;
        if (s < 0.0f || 1.0f < s) continue;
        ++hits;
        if (t < best) best = t;
    }
    ++w->stat_rays; w->stat_rays_hit += hits > 0; w->stat_rays_multi += hits > 1;
    return best;
}

/* ================================================================================================ MultiWalkerEnv */
/* constants, multi_walker.py:17-47, evaluated in float64 like the Python module does */
#define MW_FPS 50
#define MW_SCALE 30.0
#define MW_MOTORS_TORQUE 80
#define MW_SPEED_HIP 4
#define MW_SPEED_KNEE 6
#define MW_LIDAR_RANGE (160 / MW_SCALE)
#define MW_INITIAL_RANDOM 5
#define MW_LEG_DOWN (-8 / MW_SCALE)
#define MW_LEG_W (8 / MW_SCALE)
#define MW_LEG_H (34 / MW_SCALE)
#define MW_PACKAGE_LENGTH 240
#define MW_VIEWPORT_W 600
#define MW_VIEWPORT_H 400
#define MW_TERRAIN_STEP (14 / MW_SCALE)
#define MW_TERRAIN_LENGTH 200
#define MW_TERRAIN_HEIGHT (MW_VIEWPORT_H / MW_SCALE / 4)
#define MW_TERRAIN_GRASS 10
#define MW_TERRAIN_STARTPAD 20
#define MW_FRICTION 2.5
#define MW_WALKER_SEPERATION 10
#define MW_MAX_AGENTS 40
#define MW_MAX_WALKERS 10
static const double HULL_POLY[5][2] = {{-30, +9}, {+6, +9}, {+34, +1}, {+34, -8}, {-30, -8}};
static const double PACKAGE_POLY[4][2] = {{-120, 5}, {120, 5}, {120, -5}, {-120, -5}};

enum { KIND_TERRAIN = 0, KIND_PACKAGE = 1, KIND_HULL = 2, KIND_UPPER = 3, KIND_LOWER = 4 };

typedef struct {
    int32_t n_walkers, reward_global, terminate_on_fall, one_hot, continuous_physics, polygon_revision;
    double position_noise, angle_noise, forward_reward, fall_reward, drop_reward;
    uint64_t seed;
    int64_t env_id_base;
} mwr_config;

typedef struct {
    World world;
    int NT;                                   /* terrain_length (:301) */
    double terrain_y[MW_TERRAIN_LENGTH * MW_MAX_WALKERS / 8];
    int package, hull[MW_MAX_WALKERS], legs[MW_MAX_WALKERS][4], joints[MW_MAX_WALKERS][4];
    double start_x[MW_MAX_WALKERS], package_scale, package_length;
    int game_over, fallen[MW_MAX_WALKERS];
    double prev_shaping[MW_MAX_WALKERS], prev_package_shaping;
    uint32_t episode, tick;                   /* RNG counters (D3): resets of this env so far; observations of the current episode so far */
    int32_t t;
} MwEnv;

typedef struct {
    mwr_config cfg;
    int64_t n_envs;
    MwEnv *envs;
} mwr_handle;

/* Philox4x32-10 (Salmon et al., SC'11), restated from the paper; key = seed; counter = (env id, episode, index, tag) for the draws of a
 * reset and (env id, episode, observation << 6 | index, tag) for the observation noise: a reset's world is a function of the env and of
 * how many episodes it has had, not of when the previous episode ended */
static void mwr_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
enum { MWR_TAG_TERRAIN = 32, MWR_TAG_PUSH = 33, MWR_TAG_NOISE = 34 };
static inline double u24(uint32_t r) { return (double)(r >> 8) / 16777216.0; }

/* ContactDetector.BeginContact / EndContact, multi_walker.py:50-84 */
static void mw_contact_listener(void *user, World *w, int ci, int begin) {
    MwEnv *e = (MwEnv *)user;
    const Contact *c = &w->contacts[ci];
    const int bodyA = c->bodyA, bodyB = c->bodyB;
    const int W = (int)(sizeof(e->hull) / sizeof(e->hull[0]));
    (void)W;
    if (begin) {
        for (int i = 0; i < MW_MAX_WALKERS; ++i) {   /* if walkers fall on ground (:57-64) */
            if (e->hull[i] < 0) continue;
            if (e->hull[i] == bodyA && e->package != bodyB) e->fallen[i] = 1;
            if (e->hull[i] == bodyB && e->package != bodyA) e->fallen[i] = 1;
        }
        if (e->package == bodyA && w->bodies[bodyB].userKind != KIND_HULL) e->game_over = 1;   /* if package is on the ground (:66-72) */
        if (e->package == bodyB && w->bodies[bodyA].userKind != KIND_HULL) e->game_over = 1;
    }
    /* legs[1], legs[3]: the lower legs (:75-78, :81-84) */
    if (w->bodies[bodyA].userKind == KIND_LOWER) w->bodies[bodyA].userFlag = begin ? 1 : 0;
    if (w->bodies[bodyB].userKind == KIND_LOWER) w->bodies[bodyB].userFlag = begin ? 1 : 0;
}

/* MultiWalkerEnv.reset (:330-357) up to, not including, its trailing step.  terrain_in: NT heights or NULL (Philox, D3);
 * push_in: W initial pushes or NULL. */
static void mw_reset_world(const mwr_config *cfg, MwEnv *e, uint32_t gid, const double *terrain_in, const double *push_in) {
    const int W = cfg->n_walkers;
    const uint32_t k0 = (uint32_t)cfg->seed, k1 = (uint32_t)(cfg->seed >> 32);
    const uint32_t tick = e->episode;
    World *w = &e->world;
    world_init(w, V(0.0f, -10.0f));           /* Box2D.b2World(): gravity (0, -10), doSleep True (:280); a fresh world, see D1 */
    w->continuousPhysics = cfg->continuous_physics;
    w->polygonRevision = cfg->polygon_revision ? 1 : 0;
    w->listener = mw_contact_listener; w->listenerUser = e;
    e->game_over = 0; e->prev_package_shaping = 0.0; e->t = 0;
    for (int i = 0; i < MW_MAX_WALKERS; ++i) { e->fallen[i] = 0; e->prev_shaping[i] = 0.0; e->hull[i] = -1; }
    const double init_x0 = MW_TERRAIN_STEP * MW_TERRAIN_STARTPAD / 2, init_y = MW_TERRAIN_HEIGHT + 2 * MW_LEG_H;   /* :283-284 */
    double mean_x = 0.0;
    for (int i = 0; i < W; ++i) { e->start_x[i] = init_x0 + MW_WALKER_SEPERATION * i * MW_TERRAIN_STEP; mean_x += e->start_x[i]; }   /* :285-287 */
    mean_x /= W;                               /* np.mean: pairwise sum of <= 4 values = plain left-to-right sum */
    e->package_scale = W / 1.75;               /* :293 */
    e->package_length = MW_PACKAGE_LENGTH / MW_SCALE * e->package_scale;   /* :294 */
    e->NT = (int)(MW_TERRAIN_LENGTH * W * 1 / 8.);                           /* :301 */
    {   /* _generate_package (:499-514) */
        Vec2 pts[4];
        for (int k = 0; k < 4; ++k) pts[k] = V((float)(PACKAGE_POLY[k][0] * e->package_scale / MW_SCALE), (float)(PACKAGE_POLY[k][1] / MW_SCALE));
        Shape s;
        polygon_set(&s, pts, 4);
        e->package = world_create_body(w, BODY_DYNAMIC, V((float)mean_x, (float)(MW_TERRAIN_HEIGHT + 3 * MW_LEG_H)), 0.0f, &s, 1.0f, 0.5f, 0x004, 0xFFFF);
        w->bodies[e->package].userKind = KIND_PACKAGE;
    }
    {   /* _generate_terrain, hardcore == False (:516-612) */
        double velocity = 0.0, y = MW_TERRAIN_HEIGHT;
        int counter = MW_TERRAIN_STARTPAD, oneshot = 0;
        for (int i = 0; i < e->NT; ++i) {
            uint32_t r[4];
            mwr_philox(gid, tick, (uint32_t)i, MWR_TAG_TERRAIN, k0, k1, r);
            if (!oneshot) {
                const double d = MW_TERRAIN_HEIGHT - y;
                velocity = 0.8 * velocity + 0.01 * (d > 0 ? 1.0 : (d < 0 ? -1.0 : 0.0));      /* np.sign */
                if (i > MW_TERRAIN_STARTPAD) velocity += (2.0 * u24(r[0]) - 1.0) / MW_SCALE;   /* np_random.uniform(-1, 1) / SCALE */
                y += velocity;
            }
            oneshot = 0;
            e->terrain_y[i] = terrain_in ? terrain_in[i] : y;
            counter -= 1;
            if (counter == 0) {
                counter = MW_TERRAIN_GRASS / 2 + (int)(((uint64_t)r[1] * (uint64_t)(MW_TERRAIN_GRASS - MW_TERRAIN_GRASS / 2)) >> 32);   /* randint(5, 10) */
                oneshot = 1;
            }
        }
        for (int i = 0; i < e->NT - 1; ++i) {   /* :613-620 one static body per edge, friction 2.5, category 0x0001 */
            Shape s;
            memset(&s, 0, sizeof(s));
            s.type = SHAPE_EDGE; s.count = 2; s.radius = b2_polygonRadius;
            s.v[0] = V((float)(i * MW_TERRAIN_STEP), (float)e->terrain_y[i]);
            s.v[1] = V((float)((i + 1) * MW_TERRAIN_STEP), (float)e->terrain_y[i + 1]);
            const int b = world_create_body(w, BODY_STATIC, V(0, 0), 0.0f, &s, 0.0f, (float)MW_FRICTION, 0x0001, 0xFFFF);
            w->bodies[b].userKind = KIND_TERRAIN; w->bodies[b].userIndex = i;
        }
    }
    for (int wi = 0; wi < W; ++wi) {   /* BipedalWalker._reset (:113-192) */
        const double init_x = e->start_x[wi];
        Vec2 pts[5];
        for (int k = 0; k < 5; ++k) pts[k] = V((float)(HULL_POLY[k][0] / MW_SCALE), (float)(HULL_POLY[k][1] / MW_SCALE));
        Shape hs;
        polygon_set(&hs, pts, 5);
        const int hull = world_create_body(w, BODY_DYNAMIC, V((float)init_x, (float)init_y), 0.0f, &hs, 5.0f, 0.1f, 0x002, 0xFFFF);
        w->bodies[hull].userKind = KIND_HULL; w->bodies[hull].userIndex = wi;
        e->hull[wi] = hull;
        double push;
        if (push_in) push = push_in[wi];
        else { uint32_t r[4]; mwr_philox(gid, tick, (uint32_t)wi, MWR_TAG_PUSH, k0, k1, r); push = (2.0 * u24(r[0]) - 1.0) * MW_INITIAL_RANDOM; }
        w->bodies[hull].force = vadd(w->bodies[hull].force, V((float)push, 0.0f));   /* ApplyForceToCenter((uniform(-5, 5), 0), True) (:130-131) */
        for (int side = 0; side < 2; ++side) {
            const double i = side == 0 ? -1.0 : 1.0;
            Shape ls;
            polygon_set_as_box(&ls, (float)(MW_LEG_W / 2), (float)(MW_LEG_H / 2));
            const int leg = world_create_body(w, BODY_DYNAMIC, V((float)init_x, (float)(init_y - MW_LEG_H / 2 - MW_LEG_DOWN)), (float)(i * 0.05),
                                              &ls, 1.0f, 0.2f, 0x002, 0x001);
            w->bodies[leg].userKind = KIND_UPPER; w->bodies[leg].userIndex = wi;
            e->legs[wi][2 * side] = leg;
            e->joints[wi][2 * side] = world_create_revolute(w, hull, leg, V(0.0f, (float)MW_LEG_DOWN), V(0.0f, (float)(MW_LEG_H / 2)), -0.8f, 1.1f,
                                                            (float)MW_MOTORS_TORQUE, (float)i);
            polygon_set_as_box(&ls, (float)(0.8 * MW_LEG_W / 2), (float)(MW_LEG_H / 2));
            const int lower = world_create_body(w, BODY_DYNAMIC, V((float)init_x, (float)(init_y - MW_LEG_H * 3 / 2 - MW_LEG_DOWN)), (float)(i * 0.05),
                                                &ls, 1.0f, 0.2f, 0x0020, 0x001);
            w->bodies[lower].userKind = KIND_LOWER; w->bodies[lower].userIndex = wi; w->bodies[lower].userFlag = 0;   /* ground_contact = False */
            e->legs[wi][2 * side + 1] = lower;
            e->joints[wi][2 * side + 1] = world_create_revolute(w, leg, lower, V(0.0f, (float)(-MW_LEG_H / 2)), V(0.0f, (float)(MW_LEG_H / 2)), -1.6f, -0.1f,
                                                                (float)MW_MOTORS_TORQUE, 1.0f);
        }
    }
    e->episode = tick + 1;
    e->tick = 0;
}

static void joint_set_motor(World *w, int ji, float speed, float torque) {   /* b2RevoluteJoint::SetMotorSpeed / SetMaxMotorTorque */
    RevoluteJoint *j = &w->joints[ji];
    body_set_awake(&w->bodies[j->bodyA], 1); body_set_awake(&w->bodies[j->bodyB], 1);
    j->motorSpeed = speed;
    body_set_awake(&w->bodies[j->bodyA], 1); body_set_awake(&w->bodies[j->bodyB], 1);
    j->maxMotorTorque = torque;
}

/* BipedalWalker.get_observation (:205-237): 24 values */
static void mw_walker_observation(const MwEnv *e, int wi, double *state) {
    const World *w = &e->world;
    const Body *hull = &w->bodies[e->hull[wi]];
    const Vec2 pos = hull->xf.p, vel = hull->linearVelocity;
    double lidar[10];
    for (int i = 0; i < 10; ++i) {
        const Vec2 p2 = V((float)((double)pos.x + sin(1.5 * i / 10.0) * MW_LIDAR_RANGE), (float)((double)pos.y - cos(1.5 * i / 10.0) * MW_LIDAR_RANGE));
        lidar[i] = (double)world_raycast_closest(w, pos, p2, 1);
    }
    const RevoluteJoint *j0 = &w->joints[e->joints[wi][0]], *j1 = &w->joints[e->joints[wi][1]];
    const RevoluteJoint *j2 = &w->joints[e->joints[wi][2]], *j3 = &w->joints[e->joints[wi][3]];
#define JANGLE(j) ((double)(w->bodies[(j)->bodyB].sweep.a - w->bodies[(j)->bodyA].sweep.a - (j)->referenceAngle))   /* GetJointAngle, float32 */
#define JSPEED(j) ((double)(w->bodies[(j)->bodyB].angularVelocity - w->bodies[(j)->bodyA].angularVelocity))          /* GetJointSpeed */
    state[0] = (double)hull->sweep.a;
    state[1] = 2.0 * (double)hull->angularVelocity / MW_FPS;
    state[2] = 0.3 * (double)vel.x * (MW_VIEWPORT_W / MW_SCALE) / MW_FPS;
    state[3] = 0.3 * (double)vel.y * (MW_VIEWPORT_H / MW_SCALE) / MW_FPS;
    state[4] = JANGLE(j0);
    state[5] = JSPEED(j0) / MW_SPEED_HIP;
    state[6] = JANGLE(j1) + 1.0;
    state[7] = JSPEED(j1) / MW_SPEED_KNEE;
    state[8] = w->bodies[e->legs[wi][1]].userFlag ? 1.0 : 0.0;
    state[9] = JANGLE(j2);
    state[10] = JSPEED(j2) / MW_SPEED_HIP;
    state[11] = JANGLE(j3) + 1.0;
    state[12] = JSPEED(j3) / MW_SPEED_KNEE;
    state[13] = w->bodies[e->legs[wi][3]].userFlag ? 1.0 : 0.0;
    for (int i = 0; i < 10; ++i) state[14 + i] = lidar[i];
#undef JANGLE
#undef JSPEED
}

static int mw_obs_dim(const mwr_config *cfg) { return 24 + 4 + 3 + (cfg->one_hot ? MW_MAX_AGENTS : 1); }   /* :241-243 */

/* MultiWalkerEnv.step (:359-428).  obs [W][obs_dim] float64, rew [W] float64 */
static void mw_step(const mwr_config *cfg, MwEnv *e, uint32_t gid, const float *actions, double *obs, double *rew, uint8_t *done) {
    const int W = cfg->n_walkers, D = mw_obs_dim(cfg);
    World *w = &e->world;
    const uint32_t k0 = (uint32_t)cfg->seed, k1 = (uint32_t)(cfg->seed >> 32);
    for (int i = 0; i < W; ++i)   /* apply_action (:194-203) */
        for (int k = 0; k < 4; ++k) {
            const double a = (double)actions[4 * i + k];
            const double sp = (k % 2 == 0) ? MW_SPEED_HIP : MW_SPEED_KNEE;
            const double sgn = a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0);
            const double mag = fabs(a) < 0.0 ? 0.0 : (fabs(a) > 1.0 ? 1.0 : fabs(a));   /* np.clip(np.abs(a), 0, 1) */
            joint_set_motor(w, e->joints[i][k], (float)(sp * sgn), (float)(MW_MOTORS_TORQUE * mag));
        }
    world_step(w, (float)(1.0 / MW_FPS), 6 * 30, 2 * 30);   /* :365 */
    double rewards[MW_MAX_WALKERS];
    double last_x = 0.0;
    const Body *pkg = &w->bodies[e->package];
    for (int i = 0; i < W; ++i) {
        const Body *hull = &w->bodies[e->hull[i]];
        const double x = (double)hull->xf.p.x, y = (double)hull->xf.p.y;
        last_x = x;                                          /* `pos` leaks out of the loop: the LAST walker (:417, :420) */
        double *o = obs + (size_t)i * D;
        mw_walker_observation(e, i, o);
        double nz[7] = {0, 0, 0, 0, 0, 0, 0};                /* np.random.normal draws (:389-395) through the Philox contract (D3) */
        if (cfg->position_noise != 0.0 || cfg->angle_noise != 0.0) {
            for (int q = 0; q < 4; ++q) {
                uint32_t r[4];
                mwr_philox(gid, e->episode, (e->tick << 6) | (uint32_t)(i * 4 + q), MWR_TAG_NOISE, k0, k1, r);
                const double u1 = (double)((r[0] >> 8) + 1u) / 16777216.0, u2 = u24(r[1]);
                const double rad = sqrt(-2.0 * log(u1));
                nz[2 * q] = rad * cos(2.0 * 3.14159265358979323846 * u2);
                if (2 * q + 1 < 7) nz[2 * q + 1] = rad * sin(2.0 * 3.14159265358979323846 * u2);
            }
        }
        int n = 24, zi = 0;
        for (int dj = -1; dj <= 1; dj += 2) {                /* neighbours (:381-388) */
            const int j = i + dj;
            if (j < 0 || j == W) { o[n++] = 0.0; o[n++] = 0.0; }
            else {
                const double xm = ((double)w->bodies[e->hull[j]].xf.p.x - x) / e->package_length;
                const double ym = ((double)w->bodies[e->hull[j]].xf.p.y - y) / e->package_length;
                o[n++] = xm + cfg->position_noise * nz[zi++];
                o[n++] = ym + cfg->position_noise * nz[zi++];
            }
        }
        const double xd = ((double)pkg->xf.p.x - x) / e->package_length, yd = ((double)pkg->xf.p.y - y) / e->package_length;   /* :389-395 */
        o[n++] = xd + cfg->position_noise * nz[4];
        o[n++] = yd + cfg->position_noise * nz[5];
        o[n++] = (double)pkg->sweep.a + cfg->angle_noise * nz[6];
        if (cfg->one_hot) { for (int k = 0; k < MW_MAX_AGENTS; ++k) o[n++] = (k == i) ? 1.0 : 0.0; }   /* np.eye(MAX_AGENTS)[i] */
        else o[n++] = (double)i / W;
        double shaping = 0.0;                                /* :403-407 */
        shaping -= 5.0 * fabs(o[0]);
        rewards[i] = shaping - e->prev_shaping[i];
        e->prev_shaping[i] = shaping;
    }
    const double package_shaping = cfg->forward_reward * 130 * (double)pkg->xf.p.x / MW_SCALE;   /* :409-411 */
    for (int i = 0; i < W; ++i) rewards[i] += (package_shaping - e->prev_package_shaping);
    e->prev_package_shaping = package_shaping;
    int dn = 0;
    if (e->game_over || last_x < 0) { for (int i = 0; i < W; ++i) rewards[i] += cfg->drop_reward; dn = 1; }   /* :416-418 */
    if (last_x > (e->NT - MW_TERRAIN_GRASS) * MW_TERRAIN_STEP) dn = 1;                                         /* :419-420 */
    int nfallen = 0;
    for (int i = 0; i < W; ++i) { rewards[i] += cfg->fall_reward * (e->fallen[i] ? 1.0 : 0.0); nfallen += e->fallen[i]; }   /* :421 */
    if (cfg->terminate_on_fall && nfallen > 0) dn = 1;                                                                     /* :422-423 */
    if (rew) {
        if (cfg->reward_global) { double s = 0.0; for (int i = 0; i < W; ++i) s += rewards[i]; s /= W; for (int i = 0; i < W; ++i) rew[i] = s; }   /* :428 rewards.mean() */
        else for (int i = 0; i < W; ++i) rew[i] = rewards[i];
    }
    if (done) *done = (uint8_t)dn;
    e->t += 1;
    e->tick += 1;
}

/* ------------------------------------------------------------------------------------------------ C entry points (ctypes) */
int mwr_obs_dim(const mwr_config *cfg) { return mw_obs_dim(cfg); }
int mwr_uses_libm_sincos(void) {
#ifdef MWR_POLY_SINCOS
    return 0;
#else
    return 1;
#endif
}
mwr_handle *mwr_create(const mwr_config *cfg, int64_t n_envs) {
    if (cfg->n_walkers < 1 || cfg->n_walkers > MW_MAX_WALKERS) return NULL;
    mwr_handle *h = (mwr_handle *)calloc(1, sizeof(mwr_handle));
    h->cfg = *cfg; h->n_envs = n_envs;
    h->envs = (MwEnv *)calloc((size_t)n_envs, sizeof(MwEnv));
    return h;
}
void mwr_destroy(mwr_handle *h) { if (h) { free(h->envs); free(h); } }
int mwr_dims(const mwr_handle *h, int32_t *n_bodies, int32_t *n_terrain) {
    *n_bodies = 5 * h->cfg.n_walkers + 1; *n_terrain = (int)(MW_TERRAIN_LENGTH * h->cfg.n_walkers * 1 / 8.);
    return 0;
}
/* reset(mask) incl. the trailing zero-action step (:357).  terrain [N][NT] float64 / push [N][W] float64 or NULL (Philox).
 * obs [N][W][D] float64. */
void mwr_reset(mwr_handle *h, const uint8_t *mask, const double *terrain, const double *push, double *obs) {
    const int W = h->cfg.n_walkers, D = mw_obs_dim(&h->cfg);
    const int NT = (int)(MW_TERRAIN_LENGTH * W * 1 / 8.);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t n = 0; n < h->n_envs; ++n) {
        if (mask && !mask[n]) continue;
        MwEnv *e = &h->envs[n];
        const uint32_t gid = (uint32_t)(h->cfg.env_id_base + n);
        mw_reset_world(&h->cfg, e, gid, terrain ? terrain + (size_t)n * NT : NULL, push ? push + (size_t)n * W : NULL);
        float zero[4 * MW_MAX_WALKERS] = {0};
        mw_step(&h->cfg, e, gid, zero, obs + (size_t)n * W * D, NULL, NULL);
        e->t = 0;
    }
}
void mwr_step(mwr_handle *h, const float *actions, double *obs, double *rew, uint8_t *done) {
    const int W = h->cfg.n_walkers, D = mw_obs_dim(&h->cfg);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t n = 0; n < h->n_envs; ++n)
        mw_step(&h->cfg, &h->envs[n], (uint32_t)(h->cfg.env_id_base + n), actions + (size_t)n * W * 4, obs + (size_t)n * W * D, rew + (size_t)n * W, done + n);
}
static int mw_dyn_body(const MwEnv *e, int W, int k) {   /* body order of the C ABI: package, then per walker hull, upper / lower left, upper / lower right */
    if (k == 0) return e->package;
    const int wi = (k - 1) / 5, r = (k - 1) % 5;
    (void)W;
    return r == 0 ? e->hull[wi] : e->legs[wi][r - 1];
}
/* bodies float32 [N][NB][6] = centre of mass x, y, angle, vx, vy, w */
void mwr_get_bodies(const mwr_handle *h, float *out) {
    const int W = h->cfg.n_walkers, NB = 5 * W + 1;
    for (int64_t n = 0; n < h->n_envs; ++n)
        for (int k = 0; k < NB; ++k) {
            const Body *b = &h->envs[n].world.bodies[mw_dyn_body(&h->envs[n], W, k)];
            float *o = out + ((size_t)n * NB + k) * 6;
            o[0] = b->sweep.c.x; o[1] = b->sweep.c.y; o[2] = b->sweep.a; o[3] = b->linearVelocity.x; o[4] = b->linearVelocity.y; o[5] = b->angularVelocity;
        }
}
/* teacher forcing: overwrite pose and velocity of the dynamic bodies; contacts, joints, fat AABBs, sleep times stay */
void mwr_set_bodies(mwr_handle *h, const float *in) {
    const int W = h->cfg.n_walkers, NB = 5 * W + 1;
    for (int64_t n = 0; n < h->n_envs; ++n)
        for (int k = 0; k < NB; ++k) {
            Body *b = &h->envs[n].world.bodies[mw_dyn_body(&h->envs[n], W, k)];
            const float *o = in + ((size_t)n * NB + k) * 6;
            b->sweep.c = V(o[0], o[1]); b->sweep.a = o[2]; b->sweep.c0 = b->sweep.c; b->sweep.a0 = b->sweep.a;
            b->linearVelocity = V(o[3], o[4]); b->angularVelocity = o[5];
            body_sync_transform(b);
        }
}
/* joints float32 [N][NJ][6] = impulse x, y, z, motor impulse, limit state, (motor speed); aux float32 [N][NB][6] = fat AABB lo.x, lo.y, hi.x, hi.y,
 * sleep time, awake; flags uint8 [N][1 + 3 W] = game_over, fallen[W], ground_contact[W][2]; terrain float32 [N][NT] */
void mwr_get_joints(const mwr_handle *h, float *out) {
    const int W = h->cfg.n_walkers;
    for (int64_t n = 0; n < h->n_envs; ++n)
        for (int wi = 0; wi < W; ++wi)
            for (int k = 0; k < 4; ++k) {
                const RevoluteJoint *j = &h->envs[n].world.joints[h->envs[n].joints[wi][k]];
                float *o = out + ((size_t)n * 4 * W + 4 * wi + k) * 6;
                o[0] = j->impulse[0]; o[1] = j->impulse[1]; o[2] = j->impulse[2]; o[3] = j->motorImpulse; o[4] = (float)j->limitState; o[5] = j->motorSpeed;
            }
}
void mwr_get_aux(const mwr_handle *h, float *out) {
    const int W = h->cfg.n_walkers, NB = 5 * W + 1;
    for (int64_t n = 0; n < h->n_envs; ++n)
        for (int k = 0; k < NB; ++k) {
            const Body *b = &h->envs[n].world.bodies[mw_dyn_body(&h->envs[n], W, k)];
            float *o = out + ((size_t)n * NB + k) * 6;
            o[0] = b->fatAABB.lo.x; o[1] = b->fatAABB.lo.y; o[2] = b->fatAABB.hi.x; o[3] = b->fatAABB.hi.y; o[4] = b->sleepTime; o[5] = (float)b->awake;
        }
}
void mwr_get_flags(const mwr_handle *h, uint8_t *out) {
    const int W = h->cfg.n_walkers;
    for (int64_t n = 0; n < h->n_envs; ++n) {
        const MwEnv *e = &h->envs[n];
        uint8_t *o = out + (size_t)n * (1 + 3 * W);
        o[0] = (uint8_t)e->game_over;
        for (int wi = 0; wi < W; ++wi) {
            o[1 + wi] = (uint8_t)e->fallen[wi];
            o[1 + W + 2 * wi] = (uint8_t)e->world.bodies[e->legs[wi][1]].userFlag;
            o[1 + W + 2 * wi + 1] = (uint8_t)e->world.bodies[e->legs[wi][3]].userFlag;
        }
    }
}
void mwr_get_terrain(const mwr_handle *h, float *out) {
    const int NT = (int)(MW_TERRAIN_LENGTH * h->cfg.n_walkers * 1 / 8.);
    for (int64_t n = 0; n < h->n_envs; ++n)
        for (int i = 0; i < NT; ++i) out[(size_t)n * NT + i] = (float)h->envs[n].terrain_y[i];
}
/* The contact list of env n in WORLD LIST ORDER (newest first): per contact int32 [8] = body A (C-ABI order, -1 = terrain), body B,
 * terrain edge index (-1 = none), touching, point count, feature key of point 0 / 1, 0; float32 [4] = normal / tangent impulse
 * of point 0, 1.  Returns the number of contacts (at most max_contacts are written). */
int mwr_get_contacts(const mwr_handle *h, int64_t n, int32_t *ints, float *flts, int max_contacts) {
    const MwEnv *e = &h->envs[n];
    const World *w = &e->world;
    const int W = h->cfg.n_walkers, NB = 5 * W + 1;
    int abi_of[MWR_MAX_BODIES];
    for (int b = 0; b < w->bodyCount; ++b) abi_of[b] = -1;
    for (int k = 0; k < NB; ++k) abi_of[mw_dyn_body(e, W, k)] = k;
    int count = 0;
    for (int ci = w->contactList; ci >= 0; ci = w->contacts[ci].next, ++count) {
        if (count >= max_contacts) continue;
        const Contact *c = &w->contacts[ci];
        int32_t *o = ints + (size_t)count * 8;
        float *f = flts + (size_t)count * 4;
        o[0] = abi_of[c->bodyA]; o[1] = abi_of[c->bodyB];
        o[2] = w->bodies[c->bodyA].userKind == KIND_TERRAIN ? w->bodies[c->bodyA].userIndex : -1;
        o[3] = c->touching; o[4] = c->manifold.pointCount;
        o[5] = c->manifold.pointCount > 0 ? (int32_t)feature_key(c->manifold.points[0].id) : 0;
        o[6] = c->manifold.pointCount > 1 ? (int32_t)feature_key(c->manifold.points[1].id) : 0;
        o[7] = 0;   /* (m_toiCount only lives inside one SolveTOI) */
        for (int k = 0; k < 2; ++k) {
            f[2 * k] = k < c->manifold.pointCount ? c->manifold.points[k].normalImpulse : 0.0f;
            f[2 * k + 1] = k < c->manifold.pointCount ? c->manifold.points[k].tangentImpulse : 0.0f;
        }
    }
    return count;
}
/* lidar rays cast / with a hit / with hits on several terrain edges, summed over the envs */
void mwr_get_ray_stats(const mwr_handle *h, int64_t *out3) {
    out3[0] = out3[1] = out3[2] = 0;
    for (int64_t n = 0; n < h->n_envs; ++n) { out3[0] += h->envs[n].world.stat_rays; out3[1] += h->envs[n].world.stat_rays_hit; out3[2] += h->envs[n].world.stat_rays_multi; }
}
void mwr_get_stats(const mwr_handle *h, int64_t *toi_events, int64_t *contacts_created) {
    *toi_events = 0; *contacts_created = 0;
    for (int64_t n = 0; n < h->n_envs; ++n) { *toi_events += h->envs[n].world.stat_toi_events; *contacts_created += h->envs[n].world.stat_contacts_created; }
}
/* static model data for cross-checks against the product's tables: [shape 0..3 = package, hull, upper, lower][mass, I about the centre, centre x, y] */
void mwr_model(const mwr_handle *h, float *out) {
    const MwEnv *e = &h->envs[0];
    const int ids[4] = {e->package, e->hull[0], e->legs[0][0], e->legs[0][1]};
    for (int k = 0; k < 4; ++k) {
        const Body *b = &e->world.bodies[ids[k]];
        out[4 * k] = b->mass; out[4 * k + 1] = b->I; out[4 * k + 2] = b->sweep.localCenter.x; out[4 * k + 3] = b->sweep.localCenter.y;
    }
}

/* The "Hello Box2D" scene of the Box2D manual (HelloWorld.cpp): ground box 50 x 10 half-extents at (0, -10), a 1 x 1 half-extent dynamic
 * box from (0, 4), density 1, friction 0.3, Step(1/60, 6, 2) x 60.  out [steps][3] = position.x, position.y, angle after each step. */
void mwr_helloworld(float *out, int steps) {
    static __thread World w;
    world_init(&w, V(0.0f, -10.0f));
    Shape ground, box;
    polygon_set_as_box(&ground, 50.0f, 10.0f);
    polygon_set_as_box(&box, 1.0f, 1.0f);
    world_create_body(&w, BODY_STATIC, V(0.0f, -10.0f), 0.0f, &ground, 0.0f, 0.2f, 0x0001, 0xFFFF);
    const int body = world_create_body(&w, BODY_DYNAMIC, V(0.0f, 4.0f), 0.0f, &box, 1.0f, 0.3f, 0x0001, 0xFFFF);
    for (int i = 0; i < steps; ++i) {
        world_step(&w, 1.0f / 60.0f, 6, 2);
        out[3 * i] = w.bodies[body].xf.p.x; out[3 * i + 1] = w.bodies[body].xf.p.y; out[3 * i + 2] = w.bodies[body].sweep.a;
    }
}
/* ------------------------------------------------------------------------------------------------ a bare world behind a C API
 * For oracle/shims_box2d: a Python package named `Box2D` with the few classes multi_walker.py uses, whose b2World is THIS file's World.
 * With it the UNMODIFIED reference module imports and runs here -- its own reset() builds the world call by call, its own
 * apply_action / get_observation / ContactDetector / LidarCallback / reward and termination code run on top of the dynamics of this
 * restatement -- and oracle/make_golden_multiwalker.py records it (tests/golden/multiwalker_envlayer_*.npz).  That pins the ENV LAYER
 * (everything multi_walker.py itself computes, and the world it constructs) to the reference's own code; it does NOT pin the dynamics:
 * Box2D is still restated, not run (PARITY of b2World::Step UNPINNED). */
#define MWB_MAX_EVENTS 4096
typedef struct {
    World w;
    int n_events, dropped;
    int32_t events[MWB_MAX_EVENTS][3];   /* begin (1) / end (0), body A, body B -- in the order b2ContactListener would have been called */
} mwb_world;
static void mwb_listener(void *user, World *w, int ci, int begin) {
    mwb_world *m = (mwb_world *)user;
    if (m->n_events >= MWB_MAX_EVENTS) { m->dropped += 1; return; }
    int32_t *e = m->events[m->n_events++];
    e[0] = begin; e[1] = w->contacts[ci].bodyA; e[2] = w->contacts[ci].bodyB;
}
mwb_world *mwb_create(float gx, float gy, int continuous_physics) {   /* b2World(gravity, doSleep = True) */
    mwb_world *m = (mwb_world *)calloc(1, sizeof(mwb_world));
    world_init(&m->w, V(gx, gy));
    m->w.continuousPhysics = continuous_physics;
    m->w.listener = mwb_listener; m->w.listenerUser = m;
    return m;
}
void mwb_destroy(mwb_world *m) { free(m); }
/* shape_kind 0: b2PolygonShape::Set(vertices), 1: SetAsBox(verts[0], verts[1]), 2: b2EdgeShape::Set(v1, v2).  -> body id, -1: no room */
int mwb_create_body(mwb_world *m, int dynamic, float x, float y, float angle, int shape_kind, const float *verts, int n_verts, float density,
                    float friction, int category_bits, int mask_bits) {
    if (m->w.bodyCount >= MWR_MAX_BODIES) return -1;
    Shape s;
    memset(&s, 0, sizeof(s));
    if (shape_kind == 0) {
        Vec2 pts[b2_maxPolygonVertices];
        if (n_verts > b2_maxPolygonVertices) return -1;
        for (int k = 0; k < n_verts; ++k) pts[k] = V(verts[2 * k], verts[2 * k + 1]);
        polygon_set(&s, pts, n_verts);
    } else if (shape_kind == 1) polygon_set_as_box(&s, verts[0], verts[1]);
    else { s.type = SHAPE_EDGE; s.count = 2; s.radius = b2_polygonRadius; s.v[0] = V(verts[0], verts[1]); s.v[1] = V(verts[2], verts[3]); }
    return world_create_body(&m->w, dynamic ? BODY_DYNAMIC : BODY_STATIC, V(x, y), angle, &s, density, friction, (uint16_t)category_bits, (uint16_t)mask_bits);
}
/* b2RevoluteJointDef from keyword arguments (referenceAngle 0, collideConnected false); enableMotor and enableLimit must be set: this
 * restatement has no other kind */
int mwb_create_revolute(mwb_world *m, int bodyA, int bodyB, float ax, float ay, float bx, float by, float lower, float upper, float max_motor_torque,
                        float motor_speed, int enable_motor, int enable_limit) {
    if (!enable_motor || !enable_limit || m->w.jointCount >= MWR_MAX_JOINTS) return -1;
    return world_create_revolute(&m->w, bodyA, bodyB, V(ax, ay), V(bx, by), lower, upper, max_motor_torque, motor_speed);
}
void mwb_apply_force_to_center(mwb_world *m, int body, float fx, float fy, int wake) {   /* b2Body::ApplyForceToCenter */
    Body *b = &m->w.bodies[body];
    if (b->type != BODY_DYNAMIC) return;
    if (wake && !b->awake) body_set_awake(b, 1);
    if (b->awake) b->force = vadd(b->force, V(fx, fy));
}
void mwb_set_motor_speed(mwb_world *m, int joint, float speed) {   /* b2RevoluteJoint::SetMotorSpeed */
    RevoluteJoint *j = &m->w.joints[joint];
    body_set_awake(&m->w.bodies[j->bodyA], 1); body_set_awake(&m->w.bodies[j->bodyB], 1);
    j->motorSpeed = speed;
}
void mwb_set_max_motor_torque(mwb_world *m, int joint, float torque) {   /* b2RevoluteJoint::SetMaxMotorTorque */
    RevoluteJoint *j = &m->w.joints[joint];
    body_set_awake(&m->w.bodies[j->bodyA], 1); body_set_awake(&m->w.bodies[j->bodyB], 1);
    j->maxMotorTorque = torque;
}
/* b2World::Step; returns the number of Begin / EndContact calls it made (mwb_events reads them), -1 when the log overflowed */
int mwb_step(mwb_world *m, float dt, int velocity_iterations, int position_iterations) {
    m->n_events = 0; m->dropped = 0;
    world_step(&m->w, dt, velocity_iterations, position_iterations);
    return m->dropped ? -1 : m->n_events;
}
void mwb_events(const mwb_world *m, int32_t *out) { memcpy(out, m->events, sizeof(int32_t) * 3 * (size_t)m->n_events); }
/* out[10] = position (the body origin, b2Body::GetPosition) x, y, angle, linear velocity x, y, angular velocity, world centre x, y, awake, mass */
void mwb_body_state(const mwb_world *m, int body, float *out) {
    const Body *b = &m->w.bodies[body];
    out[0] = b->xf.p.x; out[1] = b->xf.p.y; out[2] = b->sweep.a; out[3] = b->linearVelocity.x; out[4] = b->linearVelocity.y; out[5] = b->angularVelocity;
    out[6] = b->sweep.c.x; out[7] = b->sweep.c.y; out[8] = (float)b->awake; out[9] = b->type == BODY_DYNAMIC ? b->mass : 0.0f;
}
/* out[4] = GetJointAngle, GetJointSpeed, motorSpeed, maxMotorTorque */
void mwb_joint_state(const mwb_world *m, int joint, float *out) {
    const RevoluteJoint *j = &m->w.joints[joint];
    const Body *bA = &m->w.bodies[j->bodyA], *bB = &m->w.bodies[j->bodyB];
    out[0] = bB->sweep.a - bA->sweep.a - j->referenceAngle; out[1] = bB->angularVelocity - bA->angularVelocity; out[2] = j->motorSpeed; out[3] = j->maxMotorTorque;
}
/* b2World::RayCast reduced to what LidarCallback accepts (D2): the CLOSEST hit among the edge fixtures whose category has a bit of
 * `category_mask`.  Returns the body hit (-1: none); out[5] = fraction, point x, y, normal x, y (b2EdgeShape::RayCast's output). */
int mwb_raycast_closest(const mwb_world *m, float x1, float y1, float x2, float y2, int category_mask, float *out) {
    const World *w = &m->w;
    const Vec2 p1w = V(x1, y1), p2w = V(x2, y2);
    float best = 1.0f;
    int hit = -1;
    for (int bi = 0; bi < w->bodyCount; ++bi) {
        const Body *b = &w->bodies[bi];
        if (b->shape.type != SHAPE_EDGE || (b->categoryBits & (uint16_t)category_mask) == 0) continue;
        const Vec2 p1 = rmulT(b->xf.q, vsub(p1w, b->xf.p)), p2 = rmulT(b->xf.q, vsub(p2w, b->xf.p));
        const Vec2 d = vsub(p2, p1);
        const Vec2 v1 = b->shape.v[0], v2 = b->shape.v[1];
        const Vec2 e = vsub(v2, v1);
        Vec2 normal = V(e.y, -e.x);
        vnormalize(&normal);
        const float numerator = vdot(normal, vsub(v1, p1)), denominator = vdot(normal, d);
        if (denominator == 0.0f) continue;
        const float t = numerator / denominator;
        if (t < 0.0f || 1.0f < t) continue;
        const Vec2 q = vadd(p1, vscale(t, d));
        const Vec2 r = vsub(v2, v1);
        const float rr = vdot(r, r);
        if (rr == 0.0f) continue;
        const float s = vdot(vsub(q, v1), r) / rr;
        if (s < 0.0f || 1.0f < s) continue;
        if (t < best) {
            best = t; hit = bi;
            const Vec2 nl = numerator > 0.0f ? V(-normal.x, -normal.y) : normal;   /* output->normal = numerator > 0 ? -R(normal) : R(normal) */
            const Vec2 nw = rmul(b->xf.q, nl);
            const Vec2 pw = vadd(p1w, vscale(t, vsub(p2w, p1w)));                  /* b2World::RayCast: point = (1 - fraction) p1 + fraction p2 */
            out[0] = t; out[1] = pw.x; out[2] = pw.y; out[3] = nw.x; out[4] = nw.y;
        }
    }
    return hit;
}
int mwb_counts(const mwb_world *m, int32_t *bodies, int32_t *joints, int32_t *contacts) {
    *bodies = m->w.bodyCount; *joints = m->w.jointCount;
    int n = 0;
    for (int ci = m->w.contactList; ci >= 0; ci = m->w.contacts[ci].next) ++n;
    *contacts = n;
    return 0;
}

int mwr_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
