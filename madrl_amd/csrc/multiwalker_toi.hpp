// multiwalker_toi.hpp -- the geometric half of Box2D's continuous-collision pass (b2World::SolveTOI), restated from
// its published structure (Box2D 2.3.x: b2Distance.cpp, b2TimeOfImpact.cpp): GJK distance between two convex proxies
// with a simplex cache, the separation function along a witness axis, and the conservative-advancement root finder
// that returns the first time in [0, 1] at which two swept shapes come within `target` of each other.
// Host/device code like multiwalker_core.hpp, which includes this file.  PARITY UNPINNED (see there).
#pragma once

namespace MW_NS {

constexpr float B2_EPSILON = 1.1920928955078125e-7f;  // FLT_EPSILON
constexpr int TOI_MAX_VERTS = 5;

// Five points with NAMED members: an array, even one only ever indexed by unrolled loops and select chains, is turned back into an
// indexed object in scratch memory by the optimizer (it folds a select of two loads into one load from a selected address before the
// array is promoted to registers); members cannot be.
struct V2x5 { V2 e0, e1, e2, e3, e4; };
static_assert(TOI_MAX_VERTS == 5, "V2x5");
// (selects on the float components, operands by value: `c ? a : b` on two struct lvalues is a select between their ADDRESSES)
MW_HD V2 sel(bool c, V2 a, V2 b) { V2 r; r.x = c ? a.x : b.x; r.y = c ? a.y : b.y; return r; }
MW_HD V2 pick(const V2x5 &a, int i) { return sel(i == 4, a.e4, sel(i == 3, a.e3, sel(i == 2, a.e2, sel(i == 1, a.e1, a.e0)))); }
struct Proxy {        // b2DistanceProxy: a convex vertex set (edge: 2 vertices, polygon: its vertices); radius is not used (useRadii = false)
    V2x5 v;
    int n;
};
MW_HD int proxy_support(const Proxy &p, V2 d) {
    int best = 0;
    float bv = dot(p.v.e0, d);
    if (1 < p.n) { const float val = dot(p.v.e1, d); if (val > bv) { best = 1; bv = val; } }
    if (2 < p.n) { const float val = dot(p.v.e2, d); if (val > bv) { best = 2; bv = val; } }
    if (3 < p.n) { const float val = dot(p.v.e3, d); if (val > bv) { best = 3; bv = val; } }
    if (4 < p.n) { const float val = dot(p.v.e4, d); if (val > bv) { best = 4; bv = val; } }
    return best;
}
MW_HD V2 proxy_vertex(const Proxy &p, int i) { return pick(p.v, i); }

struct Sweep { V2 lc, c0, c; float a0, a, alpha0; };  // b2Sweep
MW_HD Xf sweep_xf(const Sweep &s, float beta) {       // b2Sweep::GetTransform
    Xf t;
    t.p = (1.0f - beta) * s.c0 + beta * s.c;
    const float angle = (1.0f - beta) * s.a0 + beta * s.a;
    t.q = rot(angle);
    t.p = t.p - mul(t.q, s.lc);
    return t;
}
MW_HD void sweep_advance(Sweep &s, float alpha) {     // b2Sweep::Advance
    const float beta = (alpha - s.alpha0) / (1.0f - s.alpha0);
    s.c0 = s.c0 + beta * (s.c - s.c0);
    s.a0 += beta * (s.a - s.a0);
    s.alpha0 = alpha;
}
MW_HD void sweep_normalize(Sweep &s) {                // b2Sweep::Normalize
    const float two_pi = 2.0f * B2_PI;
    const float d = two_pi * floorf(s.a0 / two_pi);
    s.a0 -= d; s.a -= d;
}

struct SimplexCache { float metric; int count; int ia[3], ib[3]; };
struct SimplexVertex { V2 wA, wB, w; float a; int ia, ib; };
struct Simplex { SimplexVertex v1, v2, v3; int count; };

MW_HD float simplex_metric(const Simplex &s) {
    if (s.count == 2) { const V2 d = s.v1.w - s.v2.w; return sqrtf(dot(d, d)); }
    if (s.count == 3) return cross(s.v2.w - s.v1.w, s.v3.w - s.v1.w);
    return 0.0f;
}
MW_HD void simplex_set(SimplexVertex &v, int ia, int ib, const Proxy &pA, Xf xfA, const Proxy &pB, Xf xfB) {
    v.ia = ia; v.ib = ib;
    v.wA = mul(xfA, proxy_vertex(pA, ia)); v.wB = mul(xfB, proxy_vertex(pB, ib));
    v.w = v.wB - v.wA;
}
MW_HD void simplex_solve2(Simplex &s) {
    const V2 w1 = s.v1.w, w2 = s.v2.w, e12 = w2 - w1;
    const float d12_2 = -dot(w1, e12);
    if (d12_2 <= 0.0f) { s.v1.a = 1.0f; s.count = 1; return; }
    const float d12_1 = dot(w2, e12);
    if (d12_1 <= 0.0f) { s.v2.a = 1.0f; s.count = 1; s.v1 = s.v2; return; }
    const float inv = 1.0f / (d12_1 + d12_2);
    s.v1.a = d12_1 * inv; s.v2.a = d12_2 * inv; s.count = 2;
}
MW_HD void simplex_solve3(Simplex &s) {
    const V2 w1 = s.v1.w, w2 = s.v2.w, w3 = s.v3.w;
    const V2 e12 = w2 - w1; const float d12_1 = dot(w2, e12), d12_2 = -dot(w1, e12);
    const V2 e13 = w3 - w1; const float d13_1 = dot(w3, e13), d13_2 = -dot(w1, e13);
    const V2 e23 = w3 - w2; const float d23_1 = dot(w3, e23), d23_2 = -dot(w2, e23);
    const float n123 = cross(e12, e13);
    const float d123_1 = n123 * cross(w2, w3), d123_2 = n123 * cross(w3, w1), d123_3 = n123 * cross(w1, w2);
    if (d12_2 <= 0.0f && d13_2 <= 0.0f) { s.v1.a = 1.0f; s.count = 1; return; }
    if (d12_1 > 0.0f && d12_2 > 0.0f && d123_3 <= 0.0f) { const float inv = 1.0f / (d12_1 + d12_2); s.v1.a = d12_1 * inv; s.v2.a = d12_2 * inv; s.count = 2; return; }
    if (d13_1 > 0.0f && d13_2 > 0.0f && d123_2 <= 0.0f) { const float inv = 1.0f / (d13_1 + d13_2); s.v1.a = d13_1 * inv; s.v3.a = d13_2 * inv; s.count = 2; s.v2 = s.v3; return; }
    if (d12_1 <= 0.0f && d23_2 <= 0.0f) { s.v2.a = 1.0f; s.count = 1; s.v1 = s.v2; return; }
    if (d13_1 <= 0.0f && d23_1 <= 0.0f) { s.v3.a = 1.0f; s.count = 1; s.v1 = s.v3; return; }
    if (d23_1 > 0.0f && d23_2 > 0.0f && d123_1 <= 0.0f) { const float inv = 1.0f / (d23_1 + d23_2); s.v2.a = d23_1 * inv; s.v3.a = d23_2 * inv; s.count = 2; s.v1 = s.v3; return; }
    const float inv = 1.0f / (d123_1 + d123_2 + d123_3);
    s.v1.a = d123_1 * inv; s.v2.a = d123_2 * inv; s.v3.a = d123_3 * inv; s.count = 3;
}

// b2Distance with useRadii = false: distance between the two vertex sets; updates the cache
MW_HD float gjk_distance(SimplexCache &cache, const Proxy &pA, Xf xfA, const Proxy &pB, Xf xfB) {
    Simplex s;
    s.count = cache.count;  // ReadCache
    if (s.count >= 1) { simplex_set(s.v1, cache.ia[0], cache.ib[0], pA, xfA, pB, xfB); s.v1.a = 0.0f; }
    if (s.count >= 2) { simplex_set(s.v2, cache.ia[1], cache.ib[1], pA, xfA, pB, xfB); s.v2.a = 0.0f; }
    if (s.count >= 3) { simplex_set(s.v3, cache.ia[2], cache.ib[2], pA, xfA, pB, xfB); s.v3.a = 0.0f; }
    if (s.count > 1) {
        const float m1 = cache.metric, m2 = simplex_metric(s);
        if (m2 < 0.5f * m1 || 2.0f * m1 < m2 || m2 < B2_EPSILON) s.count = 0;
    }
    if (s.count == 0) { simplex_set(s.v1, 0, 0, pA, xfA, pB, xfB); s.v1.a = 1.0f; s.count = 1; }
    int saveA[3] = {0, 0, 0}, saveB[3] = {0, 0, 0};
    int iter = 0;
    MW_FLOPS(60);   // cache read-back / metric, witness points and distance at the end
    while (iter < 20) {
        MW_FLOPS(70);   // simplex solve, search direction, two support searches (rotations, dot products over the vertices)
        const int save_count = s.count;
        saveA[0] = s.v1.ia; saveB[0] = s.v1.ib; saveA[1] = s.v2.ia; saveB[1] = s.v2.ib; saveA[2] = s.v3.ia; saveB[2] = s.v3.ib;
        if (s.count == 2) simplex_solve2(s);
        else if (s.count == 3) simplex_solve3(s);
        if (s.count == 3) break;  // the origin is inside the triangle
        V2 d;                     // GetSearchDirection
        if (s.count == 1) d = -s.v1.w;
        else {
            const V2 e12 = s.v2.w - s.v1.w;
            const float sgn = cross(e12, -s.v1.w);
            d = sgn > 0.0f ? cross(1.0f, e12) : cross(e12, 1.0f);
        }
        if (dot(d, d) < B2_EPSILON * B2_EPSILON) break;  // the origin lies on a segment: overlapped
        SimplexVertex nv;
        simplex_set(nv, proxy_support(pA, mulT(xfA.q, -d)), proxy_support(pB, mulT(xfB.q, d)), pA, xfA, pB, xfB);
        nv.a = 0.0f;
        ++iter;
        bool duplicate = false;
        for (int i = 0; i < 3; ++i) if (i < save_count && nv.ia == saveA[i] && nv.ib == saveB[i]) duplicate = true;
        if (duplicate) break;
        if (s.count == 1) s.v2 = nv; else s.v3 = nv;
        ++s.count;
    }
    V2 pa, pb;  // GetWitnessPoints
    if (s.count == 1) { pa = s.v1.wA; pb = s.v1.wB; }
    else if (s.count == 2) { pa = s.v1.a * s.v1.wA + s.v2.a * s.v2.wA; pb = s.v1.a * s.v1.wB + s.v2.a * s.v2.wB; }
    else { pa = s.v1.a * s.v1.wA + s.v2.a * s.v2.wA + s.v3.a * s.v3.wA; pb = pa; }
    cache.metric = simplex_metric(s);  // WriteCache
    cache.count = s.count;
    cache.ia[0] = s.v1.ia; cache.ib[0] = s.v1.ib; cache.ia[1] = s.v2.ia; cache.ib[1] = s.v2.ib; cache.ia[2] = s.v3.ia; cache.ib[2] = s.v3.ib;
    const V2 dd = pa - pb;
    return sqrtf(dot(dd, dd));
}

// b2SeparationFunction
struct SepFn { int type; V2 local_point, axis; };  // type 0 points, 1 faceA, 2 faceB
MW_HD void sep_init(SepFn &f, const SimplexCache &cache, const Proxy &pA, const Sweep &sA, const Proxy &pB, const Sweep &sB, float t1) {
    const Xf xfA = sweep_xf(sA, t1), xfB = sweep_xf(sB, t1);
    if (cache.count == 1) {
        f.type = 0;
        const V2 a = mul(xfA, proxy_vertex(pA, cache.ia[0])), b = mul(xfB, proxy_vertex(pB, cache.ib[0]));
        f.axis = b - a;
        const float len = sqrtf(dot(f.axis, f.axis));
        if (len >= B2_EPSILON) f.axis = (1.0f / len) * f.axis;   // b2Vec2::Normalize
        f.local_point = v2(0, 0);
    } else if (cache.ia[0] == cache.ia[1]) {  // two points on B, one on A
        f.type = 2;
        const V2 b1 = proxy_vertex(pB, cache.ib[0]), b2 = proxy_vertex(pB, cache.ib[1]);
        f.axis = cross(b2 - b1, 1.0f);
        { const float len = sqrtf(dot(f.axis, f.axis)); if (len >= B2_EPSILON) f.axis = (1.0f / len) * f.axis; }
        const V2 normal = mul(xfB.q, f.axis);
        f.local_point = 0.5f * (b1 + b2);
        const V2 pb = mul(xfB, f.local_point), pa = mul(xfA, proxy_vertex(pA, cache.ia[0]));
        if (dot(pa - pb, normal) < 0.0f) f.axis = -f.axis;
    } else {  // two points on A
        f.type = 1;
        const V2 a1 = proxy_vertex(pA, cache.ia[0]), a2 = proxy_vertex(pA, cache.ia[1]);
        f.axis = cross(a2 - a1, 1.0f);
        { const float len = sqrtf(dot(f.axis, f.axis)); if (len >= B2_EPSILON) f.axis = (1.0f / len) * f.axis; }
        const V2 normal = mul(xfA.q, f.axis);
        f.local_point = 0.5f * (a1 + a2);
        const V2 pa = mul(xfA, f.local_point), pb = mul(xfB, proxy_vertex(pB, cache.ib[0]));
        if (dot(pb - pa, normal) < 0.0f) f.axis = -f.axis;
    }
}
// FindMinSeparation (find = true: picks the witness indices) / Evaluate (find = false: uses the given ones)
MW_HD float sep_eval(const SepFn &f, const Proxy &pA, const Sweep &sA, const Proxy &pB, const Sweep &sB, int &ia, int &ib, float t, bool find) {
    MW_FLOPS(find ? 110 : 90);   // two sweep transforms (sin / cos), the axis in world space, support search or two points
    const Xf xfA = sweep_xf(sA, t), xfB = sweep_xf(sB, t);
    if (f.type == 0) {
        if (find) { ia = proxy_support(pA, mulT(xfA.q, f.axis)); ib = proxy_support(pB, mulT(xfB.q, -f.axis)); }
        const V2 a = mul(xfA, proxy_vertex(pA, ia)), b = mul(xfB, proxy_vertex(pB, ib));
        return dot(b - a, f.axis);
    } else if (f.type == 1) {
        const V2 normal = mul(xfA.q, f.axis), a = mul(xfA, f.local_point);
        if (find) { ia = -1; ib = proxy_support(pB, mulT(xfB.q, -normal)); }
        const V2 b = mul(xfB, proxy_vertex(pB, ib));
        return dot(b - a, normal);
    } else {
        const V2 normal = mul(xfB.q, f.axis), b = mul(xfB, f.local_point);
        if (find) { ib = -1; ia = proxy_support(pA, mulT(xfA.q, -normal)); }
        const V2 a = mul(xfA, proxy_vertex(pA, ia));
        return dot(a - b, normal);
    }
}

enum { TOI_UNKNOWN = 0, TOI_FAILED = 1, TOI_OVERLAPPED = 2, TOI_TOUCHING = 3, TOI_SEPARATED = 4 };
// b2TimeOfImpact with tMax = 1; both proxies carry the polygon radius (total radius 2 * POLY_RADIUS)
MW_HD int time_of_impact(float &t_out, const Proxy &pA, Sweep sA, const Proxy &pB, Sweep sB) {
    int state = TOI_UNKNOWN;
    t_out = 1.0f;
    sweep_normalize(sA); sweep_normalize(sB);
    const float t_max = 1.0f, total_radius = 2.0f * POLY_RADIUS;
    const float target = mxf(LINEAR_SLOP, total_radius - 3.0f * LINEAR_SLOP), tolerance = 0.25f * LINEAR_SLOP;
    float t1 = 0.0f;
    int iter = 0;
    SimplexCache cache;
    cache.count = 0; cache.metric = 0.0f;
    for (int k = 0; k < 3; ++k) { cache.ia[k] = 0; cache.ib[k] = 0; }
    for (;;) {
        const float dist = gjk_distance(cache, pA, sweep_xf(sA, t1), pB, sweep_xf(sB, t1));
        if (dist <= 0.0f) { state = TOI_OVERLAPPED; t_out = 0.0f; break; }
        if (dist < target + tolerance) { state = TOI_TOUCHING; t_out = t1; break; }
        SepFn fcn;
        sep_init(fcn, cache, pA, sA, pB, sB, t1);
        bool done = false;
        float t2 = t_max;
        int push_back = 0;
        for (;;) {
            int ia = 0, ib = 0;
            float s2 = sep_eval(fcn, pA, sA, pB, sB, ia, ib, t2, true);
            if (s2 > target + tolerance) { state = TOI_SEPARATED; t_out = t_max; done = true; break; }
            if (s2 > target - tolerance) { t1 = t2; break; }
            float s1 = sep_eval(fcn, pA, sA, pB, sB, ia, ib, t1, false);
            if (s1 < target - tolerance) { state = TOI_FAILED; t_out = t1; done = true; break; }
            if (s1 <= target + tolerance) { state = TOI_TOUCHING; t_out = t1; done = true; break; }
            int root_iter = 0;
            float a1 = t1, a2 = t2;
            for (;;) {
                float t;
                if (root_iter & 1) t = a1 + (target - s1) * (a2 - a1) / (s2 - s1);  // secant
                else t = 0.5f * (a1 + a2);                                          // bisection
                ++root_iter;
                const float s = sep_eval(fcn, pA, sA, pB, sB, ia, ib, t, false);
                if (fabsf(s - target) < tolerance) { t2 = t; break; }
                if (s > target) { a1 = t; s1 = s; } else { a2 = t; s2 = s; }
                if (root_iter == 50) break;
            }
            ++push_back;
            if (push_back == 8) break;  // b2_maxPolygonVertices
        }
        ++iter;
        if (done) break;
        if (iter == 20) { state = TOI_FAILED; t_out = t1; break; }
    }
    return state;
}

}  // namespace MW_NS
