// g++ -O2 -w scripts/mw_rounds_model.cpp -o /tmp/mw_rounds && /tmp/mw_rounds
// model: what the continuous pass's event processing costs a wavefront (16 envs x 4 lanes) in velocity sweeps,
// (a) round 3: bodies dealt to lanes in Model::toi_body order, a group of four bodies per env at a time, a group costs its longest chain
// (b) rounds: every body's k-th event of the step runs in round k, a round costs its longest event
#define MW_STATS
#include "../madrl_amd/csrc/multiwalker_core.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
mw::Stats mw::g_stats;
struct StepEv { int n; unsigned char body[64]; short sw[64]; };
int main() {
    using namespace mw;
    const int W = 3, NENV = 256, T = 150;
    Model M; memset(&M, 0, sizeof(M)); build_model(M, W);
    EnvCfg C; memset(&C, 0, sizeof(C)); C.n_walkers = W; C.terminate_on_fall = 1; C.forward_reward = 1; C.fall_reward = -100; C.drop_reward = -100; C.k0 = 1;
    std::vector<World> worlds(NENV); memset(worlds.data(), 0, sizeof(World) * NENV);
    std::vector<float> obs(W * 32), rew(W), act(4 * W);
    std::vector<std::vector<StepEv>> log(T, std::vector<StepEv>(NENV));
    uint32_t lcg = 12345;
    float zero[16] = {0};
    for (int n = 0; n < NENV; ++n) {
        Scratch S; uint8_t done = 0;
        env_reset_world(M, C, worlds[n].h, cold_view(worlds[n].c), n);
        env_step(M, C, worlds[n].h, cold_view(worlds[n].c), S, SerialPar(), n, zero, obs.data(), nullptr, nullptr);
        // desynchronise episode phases: n % 60 warm-up steps
        for (int t = 0; t < 100 + n % 60 + T; ++t) {
            for (auto &a : act) { lcg = lcg * 1664525u + 1013904223u; a = (float)(lcg >> 8) / 8388608.0f - 1.0f; }
            g_stats.ev_n = 0;
            env_step(M, C, worlds[n].h, cold_view(worlds[n].c), S, SerialPar(), n, act.data(), obs.data(), rew.data(), &done);
            const int tt = t - (100 + n % 60);
            if (tt >= 0) { StepEv &e = log[tt][n]; e.n = std::min(g_stats.ev_n, 64); memcpy(e.body, g_stats.ev_body, 64); memcpy(e.sw, g_stats.ev_sweeps, 128); }
            if (done) { env_reset_world(M, C, worlds[n].h, cold_view(worlds[n].c), n); env_step(M, C, worlds[n].h, cold_view(worlds[n].c), S, SerialPar(), n, zero, obs.data(), nullptr, nullptr); }
        }
    }
    int pos_of[MAXB]; for (int k = 0; k < M.NB; ++k) pos_of[M.toi_body[k]] = k;
    const int EV_FIXED = 60;   // sweeps-equivalent of an event's fixed part (advance, contact update, island, 20 position iterations, re-search)
    double cost_now = 0, cost_rounds = 0, cost_sum = 0, nev = 0, cnt = 0, rounds_n = 0, groups_ev = 0;
    for (int t = 0; t < T; ++t)
        for (int w0 = 0; w0 + 16 <= NENV; w0 += 16) {
            long group[4] = {0, 0, 0, 0}; long round_max[16] = {0}; int nr = 0; long total = 0;
            for (int e = w0; e < w0 + 16; ++e) {
                const StepEv &s = log[t][e];
                long chain[MAXB] = {0}; int kth[MAXB] = {0};
                for (int i = 0; i < s.n; ++i) {
                    const int b = s.body[i]; const long c = s.sw[i] + EV_FIXED;
                    chain[b] += c; total += c; nev += 1;
                    const int r = kth[b]++; if (r < 16) { round_max[r] = std::max(round_max[r], c); nr = std::max(nr, r + 1); }
                }
                for (int b = 0; b < M.NB; ++b) { const int g = pos_of[b] / 4; group[g] = std::max(group[g], chain[b]); }
            }
            long now = 0; for (int g = 0; g < 4; ++g) { now += group[g]; groups_ev += group[g] > 0; }
            long rr = 0; for (int r = 0; r < nr; ++r) rr += round_max[r];
            cost_now += now; cost_rounds += rr; cost_sum += total; cnt += 1; rounds_n += nr;
        }
    printf("events per env-step %.3f; per wavefront-step: sweeps-equivalents summed over all events %.0f\n", nev / (cnt * 16), cost_sum / cnt);
    printf("  as now (4 groups, longest chain each): %.0f (groups with an event %.2f)   rounds (k-th events together): %.0f (rounds %.2f)   ratio %.2f\n",
           cost_now / cnt, groups_ev / cnt, cost_rounds / cnt, rounds_n / cnt, cost_rounds / cost_now);
    return 0;
}
