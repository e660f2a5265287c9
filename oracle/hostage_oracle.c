/*
 * hostage_oracle.c -- CPU restatement of the reference ContinuousHostageWorld environment.
 *
 * TEST INFRASTRUCTURE ONLY (parity checker); nothing in madrl_amd/ may include, link or call it.
 *
 * Compiled twice (oracle/Makefile): HW_REAL=double, prefix hw64_ -- the arithmetic type of the reference, pinned against
 * tests/golden/hostage_*.npz (outputs of the unmodified reference); HW_REAL=float, prefix hw32_ -- the same statements in
 * the arithmetic type of the HIP kernel, for long free-running comparisons.
 *
 * Reference map (file:line under /root/reference/madrl_environments/hostage.py):
 *   sensing ............ CircAgent.sensed :62-71         reset ........ ContinuousHostageWorld.reset :137-177
 *   catch rule ......... _caught :184-198                 step ......... :228-430 (phases commented inline)
 * Quirks kept (numbered like the kernel's comments):
 *   G1  sensed() tests the ray distance against the SENSING agent's radius (:68)
 *   G2  key_loc is sampled by the first reset of an env's life only and kept afterwards (:143-146)
 *   G3  the closed gate clips BOTH coordinates to [0.5 + radius, 1] and flips the clipped velocity components (:252-258)
 *   G4  saved hostages keep colliding: they are "caught" (and paid) again whenever n_coop_save rescuers touch them (:266-272)
 *   G5  hostage distances use the saved mask and the gate state from BEFORE this step's collision processing (:296, :320, :338)
 *   G6  rewards use the gate / bombed state AFTER the processing (:385-396); cr_caught_avoid (n_coop_avoid) is never used
 *   G7  a criminal's velocity flips only if BOTH coordinates left [0, 1]; positions are never clipped (:403-408)
 *   G8  ally sensing (:311-312, :343-348, :360-362) does not reach the observation: not computed
 *   G9  local rewards pay a rescuer once per kind however many objects it touched (fancy-index +=, :392-396)
 *
 * Randomness: the reference consumes `self.np_random.rand` sequentially (MT19937).  Parity runs inject the four uniforms of
 * every criminal respawn; free-running mode uses keyed Philox4x32-10 draws: counter (global env id, tick, draw index, tag),
 * uniforms (r >> 8) * 2^-24.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef HW_REAL
#define HW_REAL double
#endif
#ifndef HW_PREFIX
#define HW_PREFIX hw64_
#endif
#define HW_CAT2(a, b) a##b
#define HW_CAT(a, b) HW_CAT2(a, b)
#define HW_FN(name) HW_CAT(HW_PREFIX, name)

typedef HW_REAL real;

typedef struct {
    int32_t n_good, n_hostages, n_bad, n_coop_save, n_coop_avoid, n_sensors;
    int32_t addid, reward_global, key_fixed, max_steps;
    double radius, bad_speed, sensor_range, action_scale, save_reward, hit_reward, encounter_reward, not_saved_reward;
    double bomb_reward, bomb_radius, key_radius, control_penalty;
    double key_loc[2];
} hw_config;

typedef struct {
    hw_config cfg;
    int64_t n_envs, env_id_base;
    uint64_t seed;
    int NP;            /* particles per env: rescuers, hostages, criminals */
    real *pos, *vel;   /* [N][NP][2] */
    real *key, *bomb;  /* [N][2] */
    uint64_t *saved;   /* curr_host_saved_mask bits */
    uint8_t *flags;    /* bit0 gate_open, bit1 bombed, bit2 key sampled */
    int32_t *t;
    uint32_t *tick;
    real *sensors;     /* [K][2] */
} hw_handle;

static inline void hw_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
enum { HW_TAG_RESPAWN = 48, HW_TAG_RESET = 49 };
static inline real hw_u24(uint32_t r) { return (real)(r >> 8) * (real)(1.0 / 16777216.0); }
static inline int hw_obs_dim(const hw_config *c) { return c->n_sensors * 5 + 5 + (c->addid ? 1 : 0); } /* :19-23 */
#define HW_SQRT(x) ((sizeof(real) == 4) ? (real)sqrtf((float)(x)) : (real)sqrt((double)(x)))
static inline real hw_dist(real ax, real ay, real bx, real by) {
    real dx = ax - bx, dy = ay - by;
    return HW_SQRT(dx * dx + dy * dy);
}
static inline real hw_clip(real v, real lo, real hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* CircAgent.sensed :62-71 for one sensor and one object */
static inline real hw_sense(real sx, real sy, real px, real py, real qx, real qy, real srange, real rad2) {
    const real rx = qx - px, ry = qy - py;
    real sv = sx * rx + sy * ry;
    const real d2 = rx * rx + ry * ry;
    if ((sv < 0) || (sv > srange) || (d2 - sv * sv > rad2)) sv = (real)INFINITY; /* G1 */
    return sv;
}

/* resp: NULL or [Nc][4] injected respawn uniforms (x, y, u_vx, u_vy) of the criminals */
static void hw_step_env(hw_handle *h, int64_t n, const real *action, const real *resp, real *obs, real *rew, uint8_t *done,
                        int32_t *info) {
    const hw_config *c = &h->cfg;
    const int Nr = c->n_good, Nh = c->n_hostages, Nc = c->n_bad, K = c->n_sensors, NP = h->NP, D = hw_obs_dim(c);
    real *X = h->pos + (size_t)n * NP * 2, *V = h->vel + (size_t)n * NP * 2;
    real *XH = X + 2 * Nr, *XC = X + 2 * (Nr + Nh), *VC = V + 2 * (Nr + Nh);
    const real kx = h->key[2 * n], ky = h->key[2 * n + 1], bx = h->bomb[2 * n], by = h->bomb[2 * n + 1];
    const real rad = (real)c->radius, r_ho = (real)(c->radius * 2);
    const uint32_t k0 = (uint32_t)h->seed, k1 = (uint32_t)(h->seed >> 32), gid = (uint32_t)(h->env_id_base + n);
    int gate_open = h->flags[n] & 1, bombed = (h->flags[n] >> 1) & 1;
    uint64_t saved = h->saved[n];
    real rewards[Nr], a[Nr][2];
    for (int i = 0; i < Nr; ++i) { a[i][0] = action[2 * i] * (real)c->action_scale; a[i][1] = action[2 * i + 1] * (real)c->action_scale; rewards[i] = 0; } /* :231 */
    for (int i = 0; i < Nr; ++i) { /* :236-238 */
        V[2 * i] = V[2 * i] + a[i][0]; V[2 * i + 1] = V[2 * i + 1] + a[i][1];
        X[2 * i] = X[2 * i] + V[2 * i]; X[2 * i + 1] = X[2 * i + 1] + V[2 * i + 1];
    }
    if (c->reward_global) { /* :241-244 */
        real s = 0;
        for (int i = 0; i < Nr; ++i) { s += a[i][0] * a[i][0]; s += a[i][1] * a[i][1]; }
        for (int i = 0; i < Nr; ++i) rewards[i] += (real)c->control_penalty * s;
    } else {
        for (int i = 0; i < Nr; ++i) rewards[i] += (real)c->control_penalty * (a[i][0] * a[i][0] + a[i][1] * a[i][1]);
    }
    for (int i = 0; i < Nr; ++i) /* walls :247-252 */
        for (int q = 0; q < 2; ++q) {
            const real cl = hw_clip(X[2 * i + q], 0, 1);
            if (X[2 * i + q] != cl) V[2 * i + q] = 0;
            X[2 * i + q] = cl;
        }
    if (!gate_open) /* gate :255-260 (G3) */
        for (int i = 0; i < Nr; ++i)
            for (int q = 0; q < 2; ++q) {
                const real cl = hw_clip(X[2 * i + q], (real)(0.5 + c->radius), 1);
                if (X[2 * i + q] != cl) V[2 * i + q] *= -1;
                X[2 * i + q] = cl;
            }
    /* collisions :263-293 */
    uint8_t col_ho[Nr][Nh > 0 ? Nh : 1], col_cr[Nr][Nc > 0 ? Nc : 1], col_bo[Nr], col_ke[Nr];
    for (int i = 0; i < Nr; ++i) {
        for (int j = 0; j < Nh; ++j) col_ho[i][j] = hw_dist(X[2 * i], X[2 * i + 1], XH[2 * j], XH[2 * j + 1]) <= rad + r_ho;
        for (int j = 0; j < Nc; ++j) col_cr[i][j] = hw_dist(X[2 * i], X[2 * i + 1], XC[2 * j], XC[2 * j + 1]) <= rad + rad;
        col_bo[i] = hw_dist(X[2 * i], X[2 * i + 1], bx, by) <= rad + (real)c->bomb_radius;
        col_ke[i] = hw_dist(X[2 * i], X[2 * i + 1], kx, ky) <= rad + (real)c->key_radius;
    }
    uint8_t ho_caught[Nh > 0 ? Nh : 1], ho_enc[Nh > 0 ? Nh : 1], cr_caught[Nc > 0 ? Nc : 1];
    int n_ho_caught = 0, n_ho_enc = 0, n_cr_caught = 0, bo_caught = 0, ke_caught = 0;
    for (int j = 0; j < Nh; ++j) {
        int s = 0;
        for (int i = 0; i < Nr; ++i) s += col_ho[i][j];
        ho_caught[j] = s >= c->n_coop_save; ho_enc[j] = s >= 1;
        n_ho_caught += ho_caught[j]; n_ho_enc += ho_enc[j];
    }
    for (int j = 0; j < Nc; ++j) {
        int s = 0;
        for (int i = 0; i < Nr; ++i) s += col_cr[i][j];
        cr_caught[j] = s >= 1; n_cr_caught += cr_caught[j];
    }
    for (int i = 0; i < Nr; ++i) { bo_caught |= col_bo[i]; ke_caught |= col_ke[i]; }
    /* sensing :295-362, observation rows [cr dist | cr speed | ho dist | key dist | bomb dist] (:398-400) */
    const real rad2 = rad * rad, srange = (real)c->sensor_range;
    for (int i = 0; i < Nr; ++i) {
        real *o = obs + (size_t)i * D;
        const real px = X[2 * i], py = X[2 * i + 1];
        for (int k = 0; k < K; ++k) {
            const real sx = h->sensors[2 * k], sy = h->sensors[2 * k + 1];
            real b = (real)INFINITY; int bi = 0;
            for (int j = 0; j < Nc; ++j) { const real sv = hw_sense(sx, sy, px, py, XC[2 * j], XC[2 * j + 1], srange, rad2); if (sv < b) { b = sv; bi = j; } }
            const int fin = b < (real)INFINITY;
            o[k] = fin ? b : 0;
            o[K + k] = fin ? (sx * (VC[2 * bi] - V[2 * i]) + sy * (VC[2 * bi + 1] - V[2 * i + 1])) : 0; /* :204-226 */
            b = (real)INFINITY;
            for (int j = 0; j < Nh; ++j) {
                real sv = hw_sense(sx, sy, px, py, XH[2 * j], XH[2 * j + 1], srange, rad2);
                if ((saved >> j) & 1) sv = (real)INFINITY; /* :296 (G5) */
                if (sv < b) b = sv;
            }
            o[2 * K + k] = (gate_open && b < (real)INFINITY) ? b : 0; /* :320-322 */
            b = hw_sense(sx, sy, px, py, kx, ky, srange, rad2);
            o[3 * K + k] = (!gate_open && b < (real)INFINITY) ? b : 0; /* :338-340 */
            b = hw_sense(sx, sy, px, py, bx, by, srange, rad2);
            o[4 * K + k] = (b < (real)INFINITY) ? b : 0;
        }
    }
    /* process collisions :365-383 */
    for (int j = 0; j < Nh; ++j) if (ho_caught[j]) saved |= (1ull << j);
    const uint32_t tick = h->tick[n];
    for (int j = 0; j < Nc; ++j)
        if (cr_caught[j]) {
            real x, y, u0, u1;
            if (resp) { x = resp[4 * j]; y = resp[4 * j + 1]; u0 = resp[4 * j + 2]; u1 = resp[4 * j + 3]; }
            else { uint32_t r[4]; hw_philox(gid, tick, (uint32_t)j, HW_TAG_RESPAWN, k0, k1, r); x = hw_u24(r[0]); y = hw_u24(r[1]); u0 = hw_u24(r[2]); u1 = hw_u24(r[3]); }
            XC[2 * j] = x; XC[2 * j + 1] = y;
            VC[2 * j] = (u0 - (real)0.5) * (real)c->bad_speed; VC[2 * j + 1] = (u1 - (real)0.5) * (real)c->bad_speed;
        }
    h->tick[n] = tick + 1;
    if (bo_caught) bombed = 1;
    if (ke_caught) gate_open = 1;
    if (c->reward_global) { /* :385-389 (G6) */
        const real add = (((real)n_ho_enc * (real)c->encounter_reward * (real)gate_open + (real)n_ho_caught * (real)c->save_reward) +
                          (real)n_cr_caught * (real)c->hit_reward) + (real)bombed * (real)c->bomb_reward;
        for (int i = 0; i < Nr; ++i) rewards[i] += add;
    } else { /* :391-396 (G9) */
        for (int i = 0; i < Nr; ++i) {
            int w_ho = 0, w_enc = 0, w_cr = 0;
            for (int j = 0; j < Nh; ++j) { w_ho |= col_ho[i][j] && ho_caught[j]; w_enc |= col_ho[i][j] && ho_enc[j]; }
            for (int j = 0; j < Nc; ++j) w_cr |= col_cr[i][j] && cr_caught[j];
            if (w_ho) rewards[i] += (real)c->save_reward;
            if (w_enc) rewards[i] += (real)c->encounter_reward * (real)gate_open;
            if (w_cr) rewards[i] += (real)c->hit_reward;
            if (col_bo[i]) rewards[i] += (real)bombed * (real)c->bomb_reward;
        }
    }
    for (int j = 0; j < Nc; ++j) { /* criminals move :402-408 (G7) */
        XC[2 * j] = XC[2 * j] + VC[2 * j]; XC[2 * j + 1] = XC[2 * j + 1] + VC[2 * j + 1];
        const int outx = !(XC[2 * j] >= 0 && XC[2 * j] <= 1), outy = !(XC[2 * j + 1] >= 0 && XC[2 * j + 1] <= 1);
        if (outx && outy) { VC[2 * j] = -1 * VC[2 * j]; VC[2 * j + 1] = -1 * VC[2 * j + 1]; }
    }
    for (int i = 0; i < Nr; ++i) { /* tail of the observation :410-425 */
        real *o = obs + (size_t)i * D + 5 * K;
        int t_ho = 0, t_cr = 0;
        for (int j = 0; j < Nh; ++j) t_ho |= col_ho[i][j];
        for (int j = 0; j < Nc; ++j) t_cr |= col_cr[i][j];
        o[0] = t_ho ? 1 : 0; o[1] = t_cr ? 1 : 0; o[2] = col_ke[i] ? 1 : 0; o[3] = col_bo[i] ? 1 : 0;
        o[4] = gate_open ? 1 : 0;
        if (c->addid) o[5] = (real)(i + 1);
    }
    h->t[n] += 1; /* :427-431 */
    const uint64_t all = Nh >= 64 ? ~0ull : ((1ull << Nh) - 1ull);
    const int limit = c->max_steps > 0 ? c->max_steps : 1000;
    const int dn = bombed || ((saved & all) == all) || (h->t[n] >= limit);
    if (dn) {
        int unsaved = 0;
        for (int j = 0; j < Nh; ++j) unsaved += !((saved >> j) & 1);
        for (int i = 0; i < Nr; ++i) rewards[i] += (real)unsaved * (real)c->not_saved_reward;
    }
    h->saved[n] = saved;
    h->flags[n] = (uint8_t)((h->flags[n] & 4) | gate_open | (bombed << 1));
    if (rew) for (int i = 0; i < Nr; ++i) rew[i] = rewards[i];
    if (done) *done = (uint8_t)dn;
    if (info) { info[0] = n_ho_caught; info[1] = n_cr_caught; }
}

/* ContinuousHostageWorld.reset :137-177; draw d of the reset is Philox(gid, tick, d, RESET): key 0, rescuer i -> 1 + i,
 * hostage j -> 1 + Nr + j (x, y, clip jitter), criminal j -> 1 + Nr + Nh + j (x, y, vx, vy), bomb -> 1 + NP */
static void hw_reset_env(hw_handle *h, int64_t n, real *obs) {
    const hw_config *c = &h->cfg;
    const int Nr = c->n_good, Nh = c->n_hostages, Nc = c->n_bad, NP = h->NP;
    real *X = h->pos + (size_t)n * NP * 2, *V = h->vel + (size_t)n * NP * 2;
    const uint32_t k0 = (uint32_t)h->seed, k1 = (uint32_t)(h->seed >> 32), gid = (uint32_t)(h->env_id_base + n);
    const uint32_t tick = h->tick[n];
    uint32_t r[4];
    h->t[n] = 0;
    if (!(h->flags[n] & 4)) { /* G2 */
        if (c->key_fixed) { h->key[2 * n] = (real)c->key_loc[0]; h->key[2 * n + 1] = (real)c->key_loc[1]; }
        else { hw_philox(gid, tick, 0u, HW_TAG_RESET, k0, k1, r); h->key[2 * n] = 1 - hw_u24(r[0]) * (real)0.1; h->key[2 * n + 1] = 1 - hw_u24(r[1]) * (real)0.1; }
    }
    for (int i = 0; i < Nr; ++i) { /* :149-153 */
        hw_philox(gid, tick, (uint32_t)(1 + i), HW_TAG_RESET, k0, k1, r);
        X[2 * i] = hw_u24(r[0]); X[2 * i + 1] = hw_clip(hw_u24(r[1]), (real)0.55, (real)0.95);
        V[2 * i] = 0; V[2 * i + 1] = 0;
    }
    for (int j = 0; j < Nh; ++j) { /* :156-160 */
        hw_philox(gid, tick, (uint32_t)(1 + Nr + j), HW_TAG_RESET, k0, k1, r);
        const int p = Nr + j;
        X[2 * p] = hw_u24(r[0]); X[2 * p + 1] = hw_clip(hw_u24(r[1]), 0, (real)0.35 + hw_u24(r[2]) * (real)0.01);
        V[2 * p] = 0; V[2 * p + 1] = 0;
    }
    for (int j = 0; j < Nc; ++j) { /* :165-168 */
        hw_philox(gid, tick, (uint32_t)(1 + Nr + Nh + j), HW_TAG_RESET, k0, k1, r);
        const int p = Nr + Nh + j;
        X[2 * p] = hw_u24(r[0]); X[2 * p + 1] = hw_u24(r[1]);
        V[2 * p] = hw_u24(r[2]) * (real)c->bad_speed; V[2 * p + 1] = hw_u24(r[3]) * (real)c->bad_speed;
    }
    hw_philox(gid, tick, (uint32_t)(1 + NP), HW_TAG_RESET, k0, k1, r); /* :171 */
    h->bomb[2 * n] = hw_clip(hw_u24(r[0]), 0, (real)0.25); h->bomb[2 * n + 1] = hw_clip(hw_u24(r[1]), 0, (real)0.25);
    h->saved[n] = 0; h->flags[n] = 4; /* gate closed, not bombed, key sampled */
    h->tick[n] = tick + 1;
    real zero[2 * (Nr > 0 ? Nr : 1)];
    memset(zero, 0, sizeof(zero));
    hw_step_env(h, n, zero, NULL, obs, NULL, NULL, NULL); /* :173 */
}

/* ------------------------------------------------------------------------------------ C API */
int HW_FN(obs_dim)(const hw_config *c) { return hw_obs_dim(c); }
int HW_FN(real_size)(void) { return (int)sizeof(real); }

hw_handle *HW_FN(create)(const hw_config *cfg, const double *sensors, int64_t n_envs, uint64_t seed, int64_t env_id_base) {
    hw_handle *h = (hw_handle *)calloc(1, sizeof(hw_handle));
    h->cfg = *cfg; h->n_envs = n_envs; h->seed = seed; h->env_id_base = env_id_base;
    h->NP = cfg->n_good + cfg->n_hostages + cfg->n_bad;
    h->pos = (real *)calloc((size_t)n_envs * h->NP * 2, sizeof(real));
    h->vel = (real *)calloc((size_t)n_envs * h->NP * 2, sizeof(real));
    h->key = (real *)calloc((size_t)n_envs * 2, sizeof(real));
    h->bomb = (real *)calloc((size_t)n_envs * 2, sizeof(real));
    h->saved = (uint64_t *)calloc(n_envs, sizeof(uint64_t));
    h->flags = (uint8_t *)calloc(n_envs, 1);
    h->t = (int32_t *)calloc(n_envs, sizeof(int32_t));
    h->tick = (uint32_t *)calloc(n_envs, sizeof(uint32_t));
    h->sensors = (real *)calloc((size_t)cfg->n_sensors * 2, sizeof(real));
    for (int k = 0; k < cfg->n_sensors * 2; ++k) h->sensors[k] = (real)sensors[k];
    return h;
}

void HW_FN(destroy)(hw_handle *h) {
    if (!h) return;
    free(h->pos); free(h->vel); free(h->key); free(h->bomb); free(h->saved); free(h->flags); free(h->t); free(h->tick); free(h->sensors); free(h);
}

void HW_FN(reset)(hw_handle *h, const uint8_t *mask, real *obs) {
    const size_t orow = (size_t)h->cfg.n_good * hw_obs_dim(&h->cfg);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < h->n_envs; ++n) {
        if (mask && !mask[n]) continue;
        hw_reset_env(h, n, obs + n * orow);
    }
}

void HW_FN(step)(hw_handle *h, const real *actions, const real *resp, real *obs, real *rew, uint8_t *done, int32_t *info) {
    const int Nr = h->cfg.n_good, Nc = h->cfg.n_bad;
    const size_t orow = (size_t)Nr * hw_obs_dim(&h->cfg);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < h->n_envs; ++n)
        hw_step_env(h, n, actions + n * Nr * 2, resp ? resp + (size_t)n * Nc * 4 : NULL, obs + n * orow, rew + n * Nr, done + n, info + 2 * n);
}

void HW_FN(get_state)(const hw_handle *h, real *pos, real *vel, real *key, real *bomb, uint64_t *saved, uint8_t *flags, int32_t *t, uint32_t *tick) {
    memcpy(pos, h->pos, sizeof(real) * h->n_envs * h->NP * 2);
    memcpy(vel, h->vel, sizeof(real) * h->n_envs * h->NP * 2);
    memcpy(key, h->key, sizeof(real) * h->n_envs * 2);
    memcpy(bomb, h->bomb, sizeof(real) * h->n_envs * 2);
    memcpy(saved, h->saved, sizeof(uint64_t) * h->n_envs);
    memcpy(flags, h->flags, h->n_envs);
    memcpy(t, h->t, sizeof(int32_t) * h->n_envs);
    memcpy(tick, h->tick, sizeof(uint32_t) * h->n_envs);
}

void HW_FN(set_state)(hw_handle *h, const real *pos, const real *vel, const real *key, const real *bomb, const uint64_t *saved, const uint8_t *flags,
                      const int32_t *t, const uint32_t *tick) {
    if (pos) memcpy(h->pos, pos, sizeof(real) * h->n_envs * h->NP * 2);
    if (vel) memcpy(h->vel, vel, sizeof(real) * h->n_envs * h->NP * 2);
    if (key) memcpy(h->key, key, sizeof(real) * h->n_envs * 2);
    if (bomb) memcpy(h->bomb, bomb, sizeof(real) * h->n_envs * 2);
    if (saved) memcpy(h->saved, saved, sizeof(uint64_t) * h->n_envs);
    if (flags) memcpy(h->flags, flags, h->n_envs);
    if (t) memcpy(h->t, t, sizeof(int32_t) * h->n_envs);
    if (tick) memcpy(h->tick, tick, sizeof(uint32_t) * h->n_envs);
}
