/*
 * multiwalker_oracle.cpp -- CPU build of the MultiWalker dynamics.
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference's arithmetic for this env lives in
 * third-party Box2D (pybox2d), which cannot be imported or built in this image and for which the
 * reference holds no golden vectors.  This file is NOT the independent restatement
 * (that is oracle/multiwalker_ref.c, which also replays Box2D's published HelloWorld output): it
 * compiles the same solver source the HIP kernel uses (madrl_amd/csrc/multiwalker_core.hpp,
 * host/device code) with g++ for the CPU.  It serves two checks: the GPU *port* (LDS, lane
 * mapping, launch sequence, spare records) bit for bit against this build, and the ALGORITHM of that source against
 * multiwalker_ref.c step by step (tests/test_multiwalker_cpu.py, no GPU needed).
 *
 * Built once per capacity class of the product source (-DMW_CAPW=4 / 8 / 10 with -DMW_NLANES=4 / 8 / 16, see the Makefile), like the
 * kernels: libmadrl_mwo_c4.so / _c8.so / _c10.so export the same symbols, oracle/multiwalker.py loads the one for its n_walkers.
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../madrl_amd/csrc/multiwalker_core.hpp"

struct MwOracle {
    mw::Model M;
    mw::EnvCfg C;
    int64_t n_envs, env_id_base;
    bool rev = false;   /* run the emulated solver lanes in descending order (the schedule must not care) */
    mw::SerialPar par() const { mw::SerialPar p; p.rev = rev; return p; }
    std::vector<mw::World> worlds;
};

extern "C" {

int mwo_obs_dim(const MwOracle *o) { return mw::obs_dim_of(o->C); }
void mwo_set_one_hot(MwOracle *o, int one_hot) { o->C.one_hot = one_hot ? 1 : 0; }
void mwo_set_lane_order(MwOracle *o, int descending) { o->rev = descending != 0; }
void mwo_set_polygon_revision(MwOracle *o, int rev) { o->M.poly_rev = rev ? 1 : 0; }   /* 0: b2CollidePolygons of Box2D 2.3.0 (default), 1: of later 2.3.x */
void mwo_set_continuous(MwOracle *o, int on) { o->M.continuous = on ? 1 : 0; }  /* b2World continuousPhysics: experiments only */
int mwo_world_bytes(void) { return (int)sizeof(mw::World); }
/* layout figures the kernels' LDS blocks are computed from (multiwalker_impl.hpp k_create): sizeof(Hot), the solver's part of Scratch, Scratch without
 * the pool, sizeof(ToiWork), sizeof(Manifold), sizeof(World), sizeof(Cold) */
void mwo_sizes(int32_t *out) {
    out[0] = (int32_t)sizeof(mw::Hot); out[1] = (int32_t)offsetof(mw::Scratch, m_bA); out[2] = (int32_t)offsetof(mw::Scratch, m);
    out[3] = (int32_t)sizeof(mw::ToiWork); out[4] = (int32_t)sizeof(mw::Manifold); out[5] = (int32_t)sizeof(mw::World); out[6] = (int32_t)sizeof(mw::Cold);
}
int mwo_capacity(void) { return mw::MAX_WALKERS; }
/* byte offsets inside the world record, for tests that poke it: Hot::overflow, Hot::batch */
void mwo_hot_offsets(int32_t *out) { out[0] = (int32_t)offsetof(mw::Hot, overflow); out[1] = (int32_t)offsetof(mw::Hot, batch); }
int mwo_lanes(void) { return mw::SOLVE_LANES; }

MwOracle *mwo_create(int n_walkers, int reward_global, int terminate_on_fall, float position_noise, float angle_noise,
                     float forward_reward, float fall_reward, float drop_reward, int64_t n_envs, uint64_t seed,
                     int64_t env_id_base) {
    if (n_walkers < 1 || n_walkers > mw::MAX_WALKERS) return nullptr;
    MwOracle *o = new MwOracle();
    memset(&o->M, 0, sizeof(o->M));
    mw::build_model(o->M, n_walkers);
    memset(&o->C, 0, sizeof(o->C));
    o->C.n_walkers = n_walkers; o->C.reward_global = reward_global; o->C.terminate_on_fall = terminate_on_fall;
    o->C.position_noise = position_noise; o->C.angle_noise = angle_noise; o->C.forward_reward = forward_reward;
    o->C.fall_reward = fall_reward; o->C.drop_reward = drop_reward;
    o->C.k0 = (uint32_t)seed; o->C.k1 = (uint32_t)(seed >> 32);
    o->n_envs = n_envs; o->env_id_base = env_id_base;
    o->worlds.resize(n_envs);
    memset(o->worlds.data(), 0, sizeof(mw::World) * n_envs);
    return o;
}
void mwo_destroy(MwOracle *o) { delete o; }

void mwo_reset(MwOracle *o, const uint8_t *mask, float *obs) {
    const int W = o->M.W;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < o->n_envs; ++n) {
        if (mask && !mask[n]) continue;
        mw::Scratch S;
        float zero[4 * mw::MAX_WALKERS] = {0};
        const uint32_t gid = (uint32_t)(o->env_id_base + n);
        mw::env_reset_world(o->M, o->C, o->worlds[n].h, mw::cold_view(o->worlds[n].c), gid);
        mw::env_step(o->M, o->C, o->worlds[n].h, mw::cold_view(o->worlds[n].c), S, o->par(), gid, zero, obs + n * W * mw::obs_dim_of(o->C), nullptr, nullptr);
        o->worlds[n].h.t = 0;
    }
}

/* reset with the terrain heights [N][NT] / initial pushes [N][W] given (float64; either may be NULL = Philox) */
void mwo_reset_with(MwOracle *o, const uint8_t *mask, const double *terrain, const double *push, float *obs) {
    const int W = o->M.W, NT = o->M.NT;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < o->n_envs; ++n) {
        if (mask && !mask[n]) continue;
        mw::Scratch S;
        float zero[4 * mw::MAX_WALKERS] = {0};
        const uint32_t gid = (uint32_t)(o->env_id_base + n);
        mw::env_reset_world(o->M, o->C, o->worlds[n].h, mw::cold_view(o->worlds[n].c), gid, terrain ? terrain + n * NT : nullptr, push ? push + n * W : nullptr);
        mw::env_step(o->M, o->C, o->worlds[n].h, mw::cold_view(o->worlds[n].c), S, o->par(), gid, zero, obs + n * W * mw::obs_dim_of(o->C), nullptr, nullptr);
        o->worlds[n].h.t = 0;
    }
}

void mwo_step(MwOracle *o, const float *actions, float *obs, float *rew, uint8_t *done) {
    const int W = o->M.W;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < o->n_envs; ++n) {
        mw::Scratch S;
        mw::env_step(o->M, o->C, o->worlds[n].h, mw::cold_view(o->worlds[n].c), S, o->par(), (uint32_t)(o->env_id_base + n), actions + n * W * 4,
                     obs + n * W * mw::obs_dim_of(o->C), rew + n * W, done + n);
    }
}

/* raw world structs (same layout as the kernel's packed state) */
void mwo_get_worlds(const MwOracle *o, void *out) { memcpy(out, o->worlds.data(), sizeof(mw::World) * o->n_envs); }
void mwo_set_worlds(MwOracle *o, const void *in) { memcpy(o->worlds.data(), in, sizeof(mw::World) * o->n_envs); }

/* body kinematics for tests: [N][NB][6] = cx, cy, angle, vx, vy, w ; flags [N][2+2W] */
/* sticky Hot::overflow per env: bit 0 a contact found no room (cache slot or manifold pool), bit 1 a stale cache entry was evicted, bit 2 the
 * event log of a continuous pass was full, bit 3 the episode has had more FindNewContacts calls than a contact's 16-bit creation stamp counts --
 * all zero means every contact Box2D would have had was simulated, in Box2D's order */
void mwo_get_overflow(const MwOracle *o, uint8_t *out) { for (int64_t n = 0; n < o->n_envs; ++n) out[n] = o->worlds[n].h.overflow; }
void mwo_get_bodies(const MwOracle *o, float *out, uint8_t *flags) {
    const int NB = o->M.NB, W = o->M.W;
    for (int64_t n = 0; n < o->n_envs; ++n) {
        const mw::Hot &w = o->worlds[n].h;
        for (int b = 0; b < NB; ++b) {
            float *p = out + (n * NB + b) * 6;
            p[0] = w.b[b].c.x; p[1] = w.b[b].c.y; p[2] = w.b[b].a; p[3] = w.b[b].v.x; p[4] = w.b[b].v.y; p[5] = w.b[b].w;
        }
        if (flags) {
            uint8_t *f = flags + n * (1 + 3 * W);
            f[0] = w.game_over;
            for (int k = 0; k < W; ++k) { f[1 + k] = w.fallen[k]; f[1 + W + 2 * k] = w.ground[k][0]; f[1 + W + 2 * k + 1] = w.ground[k][1]; }
        }
    }
}
/* teacher forcing: pose and velocity of the dynamic bodies [N][NB][6]; contacts, joints, fat AABBs, sleep times stay */
void mwo_set_bodies(MwOracle *o, const float *in) {
    const int NB = o->M.NB;
    for (int64_t n = 0; n < o->n_envs; ++n)
        for (int b = 0; b < NB; ++b) {
            const float *p = in + (n * NB + b) * 6;
            mw::Body &q = o->worlds[n].h.b[b];
            q.c = mw::v2(p[0], p[1]); q.a = p[2]; q.v = mw::v2(p[3], p[4]); q.w = p[5];
        }
}
/* joints [N][NJ][6] = impulse x, y, z, motor impulse, limit state, motor speed; aux [N][NB][6] = fat AABB, sleep time, awake */
void mwo_get_joints(const MwOracle *o, float *out) {
    const int NJ = o->M.NJ;
    for (int64_t n = 0; n < o->n_envs; ++n)
        for (int j = 0; j < NJ; ++j) {
            const mw::Joint &q = o->worlds[n].c.j[j];
            float *p = out + (n * NJ + j) * 6;
            p[0] = q.ix; p[1] = q.iy; p[2] = q.iz; p[3] = q.motor_impulse; p[4] = (float)q.limit_state; p[5] = q.motor_speed;
        }
}
void mwo_get_aux(const MwOracle *o, float *out) {
    const int NB = o->M.NB;
    for (int64_t n = 0; n < o->n_envs; ++n)
        for (int b = 0; b < NB; ++b) {
            float *p = out + (n * NB + b) * 6;
            for (int k = 0; k < 4; ++k) p[k] = o->worlds[n].c.fat[b][k];
            p[4] = o->worlds[n].c.sleep_time[b]; p[5] = o->worlds[n].h.awake.test(b) ? 1.0f : 0.0f;
        }
}
/* the contacts of env n in WORLD LIST ORDER (descending key), same record as mwr_get_contacts of multiwalker_ref.c */
int mwo_get_contacts(const MwOracle *o, int64_t n, int32_t *ints, float *flts, int max_contacts) {
    const mw::Model &M = o->M;
    const mw::ColdView Cd = mw::cold_view(const_cast<mw::Cold &>(o->worlds[n].c));
    int total = 0;
    for (int s = 0; s < M.dyn_slot_base + M.n_dyn_pairs; ++s) if (Cd.slot[s].edge >= 0 && (s >= M.dyn_slot_base || s < M.slot_base[M.NB - 1] + M.slot_cap[M.NB - 1])) ++total;
    uint64_t below = ~0ull;
    int count = 0;
    for (int it = 0; it < total; ++it) {
        int best = -1; uint64_t bk = 0;
        for (int s = 0; s < M.dyn_slot_base + M.n_dyn_pairs; ++s) {
            if (Cd.slot[s].edge < 0) continue;
            const uint64_t key = mw::slot_key(M, Cd, s);
            if (key < below && (best < 0 || key > bk)) { best = s; bk = key; }
        }
        if (best < 0) break;
        below = bk;
        if (count < max_contacts) {
            const mw::Slot &sl = Cd.slot[best];
            int32_t *p = ints + (size_t)count * 8; float *f = flts + (size_t)count * 4;
            if (best >= M.dyn_slot_base) { p[0] = M.dyn_a[best - M.dyn_slot_base]; p[1] = M.dyn_b[best - M.dyn_slot_base]; p[2] = -1; }
            else { int b = 0; while (b + 1 < M.NB && best >= M.slot_base[b + 1]) ++b; p[0] = -1; p[1] = b; p[2] = sl.edge; }
            p[3] = sl.touching; p[4] = sl.npts; p[5] = sl.npts > 0 ? (int32_t)sl.id[0] : 0; p[6] = sl.npts > 1 ? (int32_t)sl.id[1] : 0; p[7] = 0;
            for (int k = 0; k < 2; ++k) { f[2 * k] = k < sl.npts ? sl.ni[k] : 0.0f; f[2 * k + 1] = k < sl.npts ? sl.ti[k] : 0.0f; }
        }
        ++count;
    }
    return count;
}
void mwo_get_terrain(const MwOracle *o, float *out) {
    for (int64_t n = 0; n < o->n_envs; ++n) memcpy(out + n * o->M.NT, o->worlds[n].c.ty, sizeof(float) * o->M.NT);
}
int mwo_num_terrain(const MwOracle *o) { return o->M.NT; }
int mwo_num_bodies(const MwOracle *o) { return o->M.NB; }
void mwo_model_masses(const MwOracle *o, float *out8) {
    for (int s = 0; s < 4; ++s) { out8[2 * s] = 1.0f / o->M.shape[s].inv_mass; out8[2 * s + 1] = 1.0f / o->M.shape[s].inv_I; }
}
}
