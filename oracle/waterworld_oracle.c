/*
 * waterworld_oracle.c -- CPU restatement of the reference MAWaterWorld environment.
 *
 * TEST INFRASTRUCTURE ONLY (parity checker + bench cpu_baseline); nothing in madrl_amd/ may
 * include, link or call it.
 *
 * Compiled twice (oracle/Makefile): WW_REAL=double, prefix ww64_ -- the arithmetic type of the
 * reference, pinned against tests/golden/waterworld_*.npz (outputs of the unmodified reference);
 * WW_REAL=float, prefix ww32_ -- the same statements in the arithmetic type of the HIP kernel
 * (north_star: "within 1e-5 for Waterworld float32 state"), used for long free-running
 * comparisons where float64-vs-float32 threshold flips would otherwise dominate.
 *
 * Reference map (file:line under /root/reference/madrl_environments/pursuit/waterworld.py):
 *   ww_sensed ........... Archea.sensed :64-72
 *   ww_reset_env ........ MAWaterWorld.reset :144-172, _respawn :139-142
 *   ww_step_env ......... MAWaterWorld.step :220-436 (phases commented inline)
 *   _caught ............. :180-193      _closest_dist :195-201     _extract_speed_features :203-218
 *
 * Randomness: the reference consumes `self.np_random.rand` sequentially (MT19937).  Parity
 * runs inject the outcome of every respawn; free-running mode uses keyed Philox4x32-10 draws
 * (DESIGN.md "RNG contract"): counter (global env id, tick, particle index, tag | attempt << 8);
 * uniforms are 24-bit (r >> 8) * 2^-24, exactly representable in float32 and float64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef WW_REAL
#define WW_REAL double
#endif
#ifndef WW_PREFIX
#define WW_PREFIX ww64_
#endif
#define WW_CAT2(a, b) a##b
#define WW_CAT(a, b) WW_CAT2(a, b)
#define WW_FN(name) WW_CAT(WW_PREFIX, name)

typedef WW_REAL real;

typedef struct {
    int32_t n_pursuers, n_evaders, n_coop, n_poison, n_sensors;
    int32_t addid, speed_features, reward_global, obstacle_fixed, max_steps;
    double radius, obstacle_radius, ev_speed, poison_speed, sensor_range, action_scale;
    double poison_reward, food_reward, encounter_reward, control_penalty;
    double obstacle_loc[2];
} ww_config;

typedef struct {
    ww_config cfg;
    int64_t n_envs, env_id_base;
    uint64_t seed;
    int NP;            /* particles per env: pursuers, evaders, poisons */
    real *pos, *vel;   /* [N][NP][2] */
    real *obst;        /* [N][2] */
    int32_t *t;        /* _timesteps */
    uint32_t *tick;    /* RNG draw counter */
    real *sensors;     /* [K][2] unit vectors (np.c_[cos, sin], :29-31) */
} ww_handle;

static inline void ww_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                             uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
enum { WW_TAG_RESPAWN = 16, WW_TAG_RESET = 17, WW_TAG_OBSTACLE = 18 };
static inline real ww_u24(uint32_t r) { return (real)(r >> 8) * (real)(1.0 / 16777216.0); }

static inline int ww_obs_dim(const ww_config *c) {
    return c->n_sensors * (c->speed_features ? 7 : 4) + 2 + (c->addid ? 1 : 0);
}

/* scipy cdist 'euclidean': sqrt(dx^2 + dy^2) */
#define WW_SQRT(x) ((sizeof(real) == 4) ? (real)sqrtf((float)(x)) : (real)sqrt((double)(x)))
static inline real ww_dist_r(real ax, real ay, real bx, real by) {
    real dx = ax - bx, dy = ay - by;
    return WW_SQRT(dx * dx + dy * dy);
}

/* respawn position for particle j: first draw + rejection loop (_respawn :139-142) */
static void ww_draw_position(const ww_handle *h, int64_t n, uint32_t tick, int j, uint32_t tag, real pradius,
                             real *ox, real *oy, real *u_vx, real *u_vy) {
    const ww_config *c = &h->cfg;
    const uint32_t k0 = (uint32_t)h->seed, k1 = (uint32_t)(h->seed >> 32), gid = (uint32_t)(h->env_id_base + n);
    const real thr = pradius * (real)2 + (real)c->obstacle_radius;
    uint32_t r[4];
    real x = 0, y = 0;
    for (uint32_t att = 0; att < 1024u; ++att) {
        ww_philox(gid, tick, (uint32_t)j, tag | (att << 8), k0, k1, r);
        x = ww_u24(r[0]);
        y = ww_u24(r[1]);
        if (att == 0) { *u_vx = ww_u24(r[2]); *u_vy = ww_u24(r[3]); }
        if (!(ww_dist_r(x, y, h->obst[2 * n], h->obst[2 * n + 1]) <= thr)) break;
    }
    *ox = x; *oy = y;
}

/* MAWaterWorld.step :220-436 for one env.
 * resp: NULL (free-running Philox) or [NP][4] injected respawn outcomes (x, y, u_vx, u_vy). */
static void ww_step_env(ww_handle *h, int64_t n, const real *action, const real *resp, real *obs, real *rew,
                        uint8_t *done, int32_t *info) {
    const ww_config *c = &h->cfg;
    const int Np = c->n_pursuers, Ne = c->n_evaders, Npo = c->n_poison, K = c->n_sensors, NP = h->NP;
    real *X = h->pos + (size_t)n * NP * 2, *V = h->vel + (size_t)n * NP * 2;
    const real ox = h->obst[2 * n], oy = h->obst[2 * n + 1];
    const real r_pu = (real)c->radius, r_ev = (real)(c->radius * 2), r_po = (real)(c->radius * 3 / 4); /* :108-118 */
    const real obst_r = (real)c->obstacle_radius;
    real rewards[Np];
    real a[Np][2];

    /* :221-231 scale actions, integrate pursuers */
    for (int i = 0; i < Np; ++i) {
        a[i][0] = action[2 * i] * (real)c->action_scale;
        a[i][1] = action[2 * i + 1] * (real)c->action_scale;
        rewards[i] = 0;
        V[2 * i] = V[2 * i] + a[i][0];
        V[2 * i + 1] = V[2 * i + 1] + a[i][1];
        X[2 * i] = X[2 * i] + V[2 * i];
        X[2 * i + 1] = X[2 * i + 1] + V[2 * i + 1];
    }
    /* :233-237 control penalty */
    if (c->reward_global) {
        real s = 0;
        for (int i = 0; i < Np; ++i) { s += a[i][0] * a[i][0]; s += a[i][1] * a[i][1]; } /* (a**2).sum(): row-major */
        for (int i = 0; i < Np; ++i) rewards[i] += (real)c->control_penalty * s;
    } else {
        for (int i = 0; i < Np; ++i) rewards[i] += (real)c->control_penalty * (a[i][0] * a[i][0] + a[i][1] * a[i][1]);
    }
    /* :239-245 walls: clip position, zero the clipped velocity components */
    for (int i = 0; i < Np; ++i)
        for (int d = 0; d < 2; ++d) {
            real x = X[2 * i + d], cl = x < 0 ? (real)0 : (x > 1 ? (real)1 : x);
            if (x != cl) V[2 * i + d] = 0;
            X[2 * i + d] = cl;
        }
    /* :247-270 obstacle rebound (velocity only) */
    for (int j = 0; j < NP; ++j) {
        const real pr = j < Np ? r_pu : (j < Np + Ne ? r_ev : r_po);
        if (ww_dist_r(X[2 * j], X[2 * j + 1], ox, oy) <= pr + obst_r) {
            const real f = (j < Np + Ne) ? (real)(-1.0 / 2) : (real)-1; /* -1/2*v (py3), poison -1*v */
            V[2 * j] = f * V[2 * j];
            V[2 * j + 1] = f * V[2 * j + 1];
        }
    }
    /* :272-293 collisions */
    uint8_t col_ev[Np][Ne > 0 ? Ne : 1], col_po[Np][Npo > 0 ? Npo : 1];
    for (int i = 0; i < Np; ++i) {
        for (int e = 0; e < Ne; ++e)
            col_ev[i][e] = ww_dist_r(X[2 * i], X[2 * i + 1], X[2 * (Np + e)], X[2 * (Np + e) + 1]) <= r_pu + r_ev;
        for (int p = 0; p < Npo; ++p)
            col_po[i][p] = ww_dist_r(X[2 * i], X[2 * i + 1], X[2 * (Np + Ne + p)], X[2 * (Np + Ne + p) + 1]) <= r_pu + r_po;
    }
    /* _caught :180-193 */
    uint8_t ev_caught[Ne > 0 ? Ne : 1], po_caught[Npo > 0 ? Npo : 1], ev_enc[Ne > 0 ? Ne : 1];
    int n_evc = 0, n_poc = 0, n_enc = 0;
    for (int e = 0; e < Ne; ++e) {
        int s = 0;
        for (int i = 0; i < Np; ++i) s += col_ev[i][e];
        ev_caught[e] = s >= c->n_coop; n_evc += ev_caught[e];
        ev_enc[e] = s >= 1; n_enc += ev_enc[e];
    }
    for (int p = 0; p < Npo; ++p) {
        int s = 0;
        for (int i = 0; i < Np; ++i) s += col_po[i][p];
        po_caught[p] = s >= 1; n_poc += po_caught[p];
    }
    /* :295-353 sensing: per (pursuer, sensor) closest object of each class along the ray */
    const int nfeat = c->speed_features ? 7 : 4;
    const real srange = (real)c->sensor_range, rad2 = r_pu * r_pu; /* W3: the SENSING pursuer's radius */
    const real INF = (real)INFINITY;
    for (int i = 0; i < Np; ++i) {
        real *o = obs + (size_t)i * ww_obs_dim(c);
        const real px = X[2 * i], py = X[2 * i + 1], pvx = V[2 * i], pvy = V[2 * i + 1];
        for (int k = 0; k < K; ++k) {
            const real sx = h->sensors[2 * k], sy = h->sensors[2 * k + 1];
            /* classes: 0 obstacle, 1 evaders, 2 poison, 3 allies */
            real best[4];
            int arg[4];
            for (int cls = 0; cls < 4; ++cls) {
                int lo, cnt;
                if (cls == 0) { lo = -1; cnt = 1; }
                else if (cls == 1) { lo = Np; cnt = Ne; }
                else if (cls == 2) { lo = Np + Ne; cnt = Npo; }
                else { lo = 0; cnt = Np; }
                real b = INF;
                int bi = 0; /* np.argmin of an all-inf row is 0 */
                for (int m = 0; m < cnt; ++m) {
                    const real qx = (cls == 0) ? ox : X[2 * (lo + m)], qy = (cls == 0) ? oy : X[2 * (lo + m) + 1];
                    const real rx = qx - px, ry = qy - py;
                    real sv = sx * rx + sy * ry; /* sensors.dot(relpos.T) :67 */
                    const real d2 = rx * rx + ry * ry;
                    if ((sv < 0) || (sv > srange) || (d2 - sv * sv > rad2)) sv = INF; /* :68-69 */
                    if (cls == 3 && m == i) sv = INF;                                  /* same=True :70-71 */
                    if (sv < b) { b = sv; bi = m; }                                   /* first minimum */
                }
                best[cls] = b;
                arg[cls] = bi;
            }
            /* distance features: raw distance or 0 (:311-334, W4) */
            const real f_ob = isfinite((double)best[0]) ? best[0] : (real)0;
            real f_d[3], f_s[3];
            for (int cls = 1; cls < 4; ++cls) {
                const int lo = cls == 1 ? Np : (cls == 2 ? Np + Ne : 0);
                const int fin = isfinite((double)best[cls]);
                f_d[cls - 1] = fin ? best[cls] : (real)0;
                /* _extract_speed_features :203-218: sensors.dot(v_obj - v_pursuer) at the closest index */
                const int j = lo + arg[cls];
                const int cnt = cls == 1 ? Ne : (cls == 2 ? Npo : Np);
                real sp = 0;
                if (fin && cnt > 0) sp = sx * (V[2 * j] - pvx) + sy * (V[2 * j + 1] - pvy);
                f_s[cls - 1] = sp;
            }
            /* np.c_[ob, evd, evs, pod, pos, pud, pus] -> blocks of K (:389-395) */
            if (c->speed_features) {
                o[0 * K + k] = f_ob; o[1 * K + k] = f_d[0]; o[2 * K + k] = f_s[0]; o[3 * K + k] = f_d[1];
                o[4 * K + k] = f_s[1]; o[5 * K + k] = f_d[2]; o[6 * K + k] = f_s[2];
            } else {
                o[0 * K + k] = f_ob; o[1 * K + k] = f_d[0]; o[2 * K + k] = f_d[1]; o[3 * K + k] = f_d[2];
            }
        }
        /* :411-428 collision flags and 1-based id (W10) */
        int tev = 0, tpo = 0;
        for (int e = 0; e < Ne; ++e) tev += col_ev[i][e];
        for (int p = 0; p < Npo; ++p) tpo += col_po[i][p];
        o[nfeat * K] = tev > 0 ? (real)1 : (real)0;
        o[nfeat * K + 1] = tpo > 0 ? (real)1 : (real)0;
        if (c->addid) o[nfeat * K + 2] = (real)(i + 1);
    }
    /* :355-374 respawn caught evaders, then caught poisons */
    const uint32_t tick = h->tick[n];
    for (int j = Np; j < NP; ++j) {
        const int is_ev = j < Np + Ne;
        if (!(is_ev ? ev_caught[j - Np] : po_caught[j - Np - Ne])) continue;
        real x, y, u0, u1;
        if (resp) {
            x = resp[4 * j]; y = resp[4 * j + 1]; u0 = resp[4 * j + 2]; u1 = resp[4 * j + 3];
        } else {
            ww_draw_position(h, n, tick, j, WW_TAG_RESPAWN, is_ev ? r_ev : r_po, &x, &y, &u0, &u1);
        }
        const real sp = (real)(is_ev ? c->ev_speed : c->poison_speed); /* W9 */
        X[2 * j] = x; X[2 * j + 1] = y;
        V[2 * j] = (u0 - (real)0.5) * sp;
        V[2 * j + 1] = (u1 - (real)0.5) * sp;
    }
    h->tick[n] = tick + 1;
    /* :376-385 rewards */
    if (c->reward_global) {
        const real add = ((real)n_evc * (real)c->food_reward) + ((real)n_poc * (real)c->poison_reward) +
                         ((real)n_enc * (real)c->encounter_reward);
        for (int i = 0; i < Np; ++i) rewards[i] += add;
    } else { /* fancy-index += : a pursuer in several simultaneous catches is paid once (W7) */
        for (int i = 0; i < Np; ++i) {
            int wc = 0, wp = 0, we = 0;
            for (int e = 0; e < Ne; ++e) { wc |= (col_ev[i][e] && ev_caught[e]); we |= (col_ev[i][e] && ev_enc[e]); }
            for (int p = 0; p < Npo; ++p) wp |= (col_po[i][p] && po_caught[p]);
            if (wc) rewards[i] += (real)c->food_reward;
            if (wp) rewards[i] += (real)c->poison_reward;
            if (we) rewards[i] += (real)c->encounter_reward;
        }
    }
    /* :397-409 evader / poison motion; bounce only if BOTH coordinates left [0,1] (W6) */
    for (int j = Np; j < NP; ++j) {
        X[2 * j] = X[2 * j] + V[2 * j];
        X[2 * j + 1] = X[2 * j + 1] + V[2 * j + 1];
        const int outx = !(X[2 * j] >= 0 && X[2 * j] <= 1), outy = !(X[2 * j + 1] >= 0 && X[2 * j + 1] <= 1);
        if (outx && outy) { V[2 * j] = (real)-1 * V[2 * j]; V[2 * j + 1] = (real)-1 * V[2 * j + 1]; }
    }
    h->t[n] += 1; /* :433 */
    if (rew) for (int i = 0; i < Np; ++i) rew[i] = rewards[i];
    if (done) *done = (uint8_t)(h->t[n] >= (c->max_steps > 0 ? c->max_steps : 1000)); /* :174-178, timestep_limit :124-126 */
    if (info) { info[0] = n_evc; info[1] = n_poc; }
}

/* MAWaterWorld.reset :144-172 (free-running draws), ends with step(zeros) (W11) */
static void ww_reset_env(ww_handle *h, int64_t n, real *obs) {
    const ww_config *c = &h->cfg;
    const int Np = c->n_pursuers, Ne = c->n_evaders, NP = h->NP;
    const uint32_t k0 = (uint32_t)h->seed, k1 = (uint32_t)(h->seed >> 32), gid = (uint32_t)(h->env_id_base + n);
    real *X = h->pos + (size_t)n * NP * 2, *V = h->vel + (size_t)n * NP * 2;
    const uint32_t tick = h->tick[n];
    uint32_t r[4];
    h->t[n] = 0;
    if (c->obstacle_fixed) {
        h->obst[2 * n] = (real)c->obstacle_loc[0];
        h->obst[2 * n + 1] = (real)c->obstacle_loc[1];
    } else { /* :147-148 */
        ww_philox(gid, tick, 0u, WW_TAG_OBSTACLE, k0, k1, r);
        h->obst[2 * n] = ww_u24(r[0]);
        h->obst[2 * n + 1] = ww_u24(r[1]);
    }
    for (int j = 0; j < NP; ++j) {
        const real pr = j < Np ? (real)c->radius : (j < Np + Ne ? (real)(c->radius * 2) : (real)(c->radius * 3 / 4));
        real x, y, u0, u1;
        ww_draw_position(h, n, tick, j, WW_TAG_RESET, pr, &x, &y, &u0, &u1);
        X[2 * j] = x; X[2 * j + 1] = y;
        if (j < Np) { V[2 * j] = 0; V[2 * j + 1] = 0; }
        else { /* :164, :170 both use ev_speed (W9) */
            V[2 * j] = (u0 - (real)0.5) * (real)c->ev_speed;
            V[2 * j + 1] = (u1 - (real)0.5) * (real)c->ev_speed;
        }
    }
    h->tick[n] = tick + 1;
    real zero[2 * (Np > 0 ? Np : 1)];
    memset(zero, 0, sizeof(zero));
    ww_step_env(h, n, zero, NULL, obs, NULL, NULL, NULL);
}

/* ------------------------------------------------------------------------------------ C API */
int WW_FN(obs_dim)(const ww_config *c) { return ww_obs_dim(c); }
int WW_FN(real_size)(void) { return (int)sizeof(real); }

ww_handle *WW_FN(create)(const ww_config *cfg, const double *sensors, int64_t n_envs, uint64_t seed,
                         int64_t env_id_base) {
    ww_handle *h = (ww_handle *)calloc(1, sizeof(ww_handle));
    h->cfg = *cfg;
    h->n_envs = n_envs; h->seed = seed; h->env_id_base = env_id_base;
    h->NP = cfg->n_pursuers + cfg->n_evaders + cfg->n_poison;
    h->pos = (real *)calloc((size_t)n_envs * h->NP * 2, sizeof(real));
    h->vel = (real *)calloc((size_t)n_envs * h->NP * 2, sizeof(real));
    h->obst = (real *)calloc((size_t)n_envs * 2, sizeof(real));
    h->t = (int32_t *)calloc(n_envs, sizeof(int32_t));
    h->tick = (uint32_t *)calloc(n_envs, sizeof(uint32_t));
    h->sensors = (real *)calloc((size_t)cfg->n_sensors * 2, sizeof(real));
    for (int k = 0; k < cfg->n_sensors * 2; ++k) h->sensors[k] = (real)sensors[k];
    for (int64_t n = 0; n < n_envs; ++n) {
        h->obst[2 * n] = (real)cfg->obstacle_loc[0];
        h->obst[2 * n + 1] = (real)cfg->obstacle_loc[1];
    }
    return h;
}

void WW_FN(destroy)(ww_handle *h) {
    if (!h) return;
    free(h->pos); free(h->vel); free(h->obst); free(h->t); free(h->tick); free(h->sensors); free(h);
}

void WW_FN(reset)(ww_handle *h, const uint8_t *mask, real *obs) {
    const size_t orow = (size_t)h->cfg.n_pursuers * ww_obs_dim(&h->cfg);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < h->n_envs; ++n) {
        if (mask && !mask[n]) continue;
        ww_reset_env(h, n, obs + n * orow);
    }
}

void WW_FN(step)(ww_handle *h, const real *actions, const real *resp, real *obs, real *rew, uint8_t *done,
                 int32_t *info) {
    const int Np = h->cfg.n_pursuers;
    const size_t orow = (size_t)Np * ww_obs_dim(&h->cfg);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < h->n_envs; ++n)
        ww_step_env(h, n, actions + n * Np * 2, resp ? resp + (size_t)n * h->NP * 4 : NULL, obs + n * orow,
                    rew + n * Np, done + n, info + 2 * n);
}

void WW_FN(get_state)(const ww_handle *h, real *pos, real *vel, real *obst, int32_t *t, uint32_t *tick) {
    memcpy(pos, h->pos, sizeof(real) * h->n_envs * h->NP * 2);
    memcpy(vel, h->vel, sizeof(real) * h->n_envs * h->NP * 2);
    memcpy(obst, h->obst, sizeof(real) * h->n_envs * 2);
    memcpy(t, h->t, sizeof(int32_t) * h->n_envs);
    memcpy(tick, h->tick, sizeof(uint32_t) * h->n_envs);
}

void WW_FN(set_state)(ww_handle *h, const real *pos, const real *vel, const real *obst, const int32_t *t,
                      const uint32_t *tick) {
    if (pos) memcpy(h->pos, pos, sizeof(real) * h->n_envs * h->NP * 2);
    if (vel) memcpy(h->vel, vel, sizeof(real) * h->n_envs * h->NP * 2);
    if (obst) memcpy(h->obst, obst, sizeof(real) * h->n_envs * 2);
    if (t) memcpy(h->t, t, sizeof(int32_t) * h->n_envs);
    if (tick) memcpy(h->tick, tick, sizeof(uint32_t) * h->n_envs);
}
