// Times nd_build_plan (the direct solver's symbolic phase) on kNN graphs of a2's shape, single-threaded and with the first two levels of the
// dissection on threads of their own (par_min), and checks that the two plans are the same:  g++ -O3 -std=c++17 -Inr-slam_amd/csrc -Iinclude tools/micro/plan_probe.cpp -o tools/micro/bin/plan_probe -lpthread
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "nrs_nd_plan.hpp"
int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 4221, knn = argc > 2 ? atoi(argv[2]) : 16;
    std::mt19937 g(5); std::uniform_real_distribution<double> ux(-20, 20), uy(-15, 15); std::normal_distribution<double> nz(0, 1);
    std::vector<double> pos(3 * (n + 2), 0.0);
    for (int i = 0; i < n; ++i) { pos[3 * i] = ux(g); pos[3 * i + 1] = uy(g); pos[3 * i + 2] = 60 + nz(g); }
    std::vector<std::pair<int,int>> pr;
    for (int i = 0; i < n; ++i) {
        std::vector<std::pair<double,int>> d;
        for (int j = 0; j < n; ++j) if (j != i) { double dx = pos[3*i]-pos[3*j], dy = pos[3*i+1]-pos[3*j+1]; d.push_back({dx*dx+dy*dy, j}); }
        std::partial_sort(d.begin(), d.begin() + knn, d.end());
        for (int k = 0; k < knn; ++k) pr.push_back({std::min(i, d[k].second), std::max(i, d[k].second)});
    }
    std::sort(pr.begin(), pr.end()); pr.erase(std::unique(pr.begin(), pr.end()), pr.end());
    std::vector<int> pairs;
    for (auto& p : pr) { pairs.push_back(p.first); pairs.push_back(p.second); }
    for (int i = 0; i < n; ++i) for (int h = 0; h < 2; ++h) { pairs.push_back(n + h); pairs.push_back(i); }
    pairs.push_back(n + 1); pairs.push_back(n);
    std::vector<uint8_t> last(n + 2, 0); last[n] = last[n + 1] = 1;
    nrs::NdPlan P[2]; std::string err;
    for (int par = 0; par < 2; ++par) {
        double best = 1e9;
        for (int r = 0; r < 10; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            bool ok = nrs::nd_build_plan(n + 2, pos.data(), last.data(), (int)pairs.size() / 2, pairs.data(), P[par], &err, nrs::ND_LEAFN, nrs::ND_SMAXN, false, par ? 500 : 0);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (!ok) { printf("fail %s\n", err.c_str()); return 1; }
            best = std::min(best, ms);
        }
        printf("par %d: n %d pairs %zu fronts %d levels %d wgs %zu: plan %.2f ms\n", par, n, pairs.size() / 2, P[par].n_fronts, P[par].n_levels, P[par].wg.size() / 3, best);
    }
    bool same = P[0].own == P[1].own && P[0].bnd == P[1].bnd && P[0].wg == P[1].wg && P[0].elim == P[1].elim && P[0].seg == P[1].seg && P[0].pmap == P[1].pmap && P[0].lvl_fronts == P[1].lvl_fronts && P[0].ent.size() == P[1].ent.size() && memcmp(P[0].ent.data(), P[1].ent.data(), sizeof(nrs::NdEnt) * P[0].ent.size()) == 0;
    printf("plans identical: %d\n", (int)same);
    return same ? 0 : 2;
}
