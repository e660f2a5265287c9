/*
 * multiwalker_oracle.cpp -- CPU build of the MultiWalker dynamics.
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference's arithmetic for this env lives in
 * third-party Box2D (pybox2d), which cannot be imported or built in this image and for which the
 * reference holds no golden vectors; the only published anchor, Box2D's HelloWorld output,
 * is checked by box2d_kat.cpp.  Unlike the Pursuit / Waterworld oracles this
 * file is therefore NOT an independent restatement pinned to the reference: it compiles the same
 * solver source the HIP kernel uses (madrl_amd/csrc/multiwalker_core.hpp, host/device code) with
 * g++ for the CPU.  What it checks is the GPU *port* (LDS staging, lane mapping, device math
 * library) step by step; the algorithm itself is covered by physical-invariant tests.
 */
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../madrl_amd/csrc/multiwalker_core.hpp"

struct MwOracle {
    mw::Model M;
    mw::EnvCfg C;
    int64_t n_envs, env_id_base;
    std::vector<mw::World> worlds;
};

extern "C" {

int mwo_obs_dim(const MwOracle *o) { return mw::obs_dim_of(o->C); }
void mwo_set_one_hot(MwOracle *o, int one_hot) { o->C.one_hot = one_hot ? 1 : 0; }
void mwo_set_continuous(MwOracle *o, int on) { o->M.continuous = on ? 1 : 0; }  /* b2World continuousPhysics: experiments only */
int mwo_world_bytes(void) { return (int)sizeof(mw::World); }

MwOracle *mwo_create(int n_walkers, int reward_global, int terminate_on_fall, float position_noise, float angle_noise,
                     float forward_reward, float fall_reward, float drop_reward, int64_t n_envs, uint64_t seed,
                     int64_t env_id_base) {
    MwOracle *o = new MwOracle();
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
        mw::env_reset_world(o->M, o->C, o->worlds[n].h, o->worlds[n].c, gid);
        mw::env_step(o->M, o->C, o->worlds[n].h, o->worlds[n].c, S, mw::SerialPar(), gid, zero, obs + n * W * mw::obs_dim_of(o->C), nullptr, nullptr);
        o->worlds[n].h.t = 0;
    }
}

void mwo_step(MwOracle *o, const float *actions, float *obs, float *rew, uint8_t *done) {
    const int W = o->M.W;
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < o->n_envs; ++n) {
        mw::Scratch S;
        mw::env_step(o->M, o->C, o->worlds[n].h, o->worlds[n].c, S, mw::SerialPar(), (uint32_t)(o->env_id_base + n), actions + n * W * 4,
                     obs + n * W * mw::obs_dim_of(o->C), rew + n * W, done + n);
    }
}

/* raw world structs (same layout as the kernel's packed state) */
void mwo_get_worlds(const MwOracle *o, void *out) { memcpy(out, o->worlds.data(), sizeof(mw::World) * o->n_envs); }
void mwo_set_worlds(MwOracle *o, const void *in) { memcpy(o->worlds.data(), in, sizeof(mw::World) * o->n_envs); }

/* body kinematics for tests: [N][NB][6] = cx, cy, angle, vx, vy, w ; flags [N][2+2W] */
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
void mwo_get_terrain(const MwOracle *o, float *out) {
    for (int64_t n = 0; n < o->n_envs; ++n) memcpy(out + n * o->M.NT, o->worlds[n].c.ty, sizeof(float) * o->M.NT);
}
int mwo_num_terrain(const MwOracle *o) { return o->M.NT; }
int mwo_num_bodies(const MwOracle *o) { return o->M.NB; }
void mwo_model_masses(const MwOracle *o, float *out8) {
    for (int s = 0; s < 4; ++s) { out8[2 * s] = 1.0f / o->M.shape[s].inv_mass; out8[2 * s + 1] = 1.0f / o->M.shape[s].inv_I; }
}
}
