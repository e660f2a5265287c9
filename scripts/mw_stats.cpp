// How much work a MultiWalker step really is (CPU build of the solver source with counters): sub-slots of a contact sweep,
// position iterations until the early exit, active manifolds.   g++ -O2 -DMW_STATS scripts/mw_stats.cpp -o /tmp/mw_stats && /tmp/mw_stats
#define MW_STATS
#include "../madrl_amd/csrc/multiwalker_core.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
mw::Stats mw::g_stats;
int main() {
    using namespace mw;
    for (int W = 1; W <= MAX_WALKERS; ++W) {
        Model M; memset(&M, 0, sizeof(M)); build_model(M, W);
        EnvCfg C; memset(&C, 0, sizeof(C)); C.n_walkers = W; C.terminate_on_fall = 1; C.forward_reward = 1; C.fall_reward = -100; C.drop_reward = -100; C.k0 = 1;
        std::vector<World> worlds(64); memset(worlds.data(), 0, sizeof(World) * 64);
        std::vector<float> obs(W * 32), rew(W), act(4 * W);
        memset(&g_stats, 0, sizeof(g_stats));
        uint32_t lcg = 12345;
        for (int n = 0; n < 64; ++n) {
            Scratch S; uint8_t done = 0; float zero[4 * MAX_WALKERS] = {0};
            env_reset_world(M, C, worlds[n].h, cold_view(worlds[n].c), n);
            env_step(M, C, worlds[n].h, cold_view(worlds[n].c), S, SerialPar(), n, zero, obs.data(), nullptr, nullptr);
            memset(&g_stats, 0, 0);
            for (int t = 0; t < 300; ++t) {
                for (auto &a : act) { lcg = lcg * 1664525u + 1013904223u; a = (float)(lcg >> 8) / 8388608.0f - 1.0f; }
                env_step(M, C, worlds[n].h, cold_view(worlds[n].c), S, SerialPar(), n, act.data(), obs.data(), rew.data(), &done);
                if (done) { env_reset_world(M, C, worlds[n].h, cold_view(worlds[n].c), n); env_step(M, C, worlds[n].h, cold_view(worlds[n].c), S, SerialPar(), n, zero, obs.data(), nullptr, nullptr); }
            }
        }
        const double s = (double)g_stats.steps;
        printf("W=%d steps=%ld  subslots A %.2f B %.2f  manifolds %.2f  merged %.2f  position iterations %.2f | TOI per step: full %.2f culled %.2f events %.3f undone %.3f velocity sweeps per event %.1f\n", W, g_stats.steps,
               g_stats.sub_a / s, g_stats.sub_b / s, g_stats.manifolds / s, g_stats.merged / s, g_stats.pos_iters / s,
               g_stats.toi_full / s, g_stats.toi_culled / s, g_stats.toi_events / s, g_stats.toi_undone / s, g_stats.toi_events ? (double)g_stats.toi_vel_iters / g_stats.toi_events : 0.0);
        printf("    sweeps until the fixed point, buckets of 20 (last = never): ");
        for (int i = 0; i < 10; ++i) printf("%ld ", g_stats.toi_hist[i]);
        printf(" | island manifolds 0..5+: ");
        for (int i = 0; i < 6; ++i) printf("%ld ", g_stats.toi_nisl[i]);
        printf("\n    largest lane list (0..23+): ");
        for (int i = 0; i < 24; ++i) printf("%ld ", g_stats.cnt_hist[i]);
        printf("\n    continuous pass: merges with several bodies %ld, ties between bodies %ld, package / hull events %ld, pairs created by the merge %ld", g_stats.toi_multi, g_stats.toi_ties, g_stats.toi_hullpkg, g_stats.toi_pairs);
        printf("\n    counted arithmetic: %.0f float operations per env-step (hand counts per primitive x measured calls: 180 + up to 60 solver sweeps, sub-step sweeps, GJK / root finder, narrow phase, lidar)", g_stats.flops / s);
        printf("\n      by phase: collide %.0f | solve %.0f | broad phase %.0f | time-of-impact search %.0f | sub-steps %.0f | observation %.0f", g_stats.flops_by[0] / s, g_stats.flops_by[1] / s,
               g_stats.flops_by[2] / s, g_stats.flops_by[3] / s, g_stats.flops_by[4] / s, g_stats.flops_by[5] / s);
        printf("\n    rounds (0..11+): ");
        for (int i = 0; i < 12; ++i) printf("%ld ", g_stats.rounds_hist[i]);
        printf("\n");
    }
}
