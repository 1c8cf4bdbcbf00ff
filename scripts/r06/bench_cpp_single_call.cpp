// What a C++ drop-in caller pays per problem through include/loik_amd/loik.hpp (no Python in the way): the reference's three entry points on ONE
// Talos-32 problem with the headline's parameters -- Solve() on a resident problem (the reference's own timing test, tests/loik-loid.cpp:987-1032),
// the full Solve(q, H_ref, v_ref, ids, Ais, bis, lb, ub) (hpp:475-580) and the tailored Solve(q, c_id, Ai, bi) (:596-695) -- each INCLUDING what the
// mirror does after a solve (the data object's z, nu, w, vis, fis, yis and the getters' scalars fetched), then get_iter() / get_convergence_status().
//   g++ -std=c++17 -O2 -I include scripts/r06/bench_cpp_single_call.cpp -o /tmp/bench_cpp -L loik_amd/lib -lloik_amd -Wl,-rpath,$PWD/loik_amd/lib -Wl,-rpath,/opt/rocm/lib
#include "loik_amd/loik.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

using namespace loik_amd;
using clk = std::chrono::steady_clock;

int main(int argc, char** argv)
{
  Model model = Model::Builtin("talos32");
  const int max_iter = 1000;
  IkIdDataOptimized data(model, 1, 1);
  FirstOrderLoikOptimized solver{max_iter, 1e-6, 0.0, 1e-2, 1e-2, 1e-5, 1e-2, 1e4, DEFAULT, 1, 6, model, data, false, 1e-1, false, false};
  DVec q(model.nq, 0.0), lb(model.nv, -0.5), ub(model.nv, 0.5);
  Vec6 b{};
  {  // the problem of bench.py's single_call_variant: scripts/r06/bench_cpp_single_call_input.txt (argv[1])
    FILE* f = std::fopen(argc > 1 ? argv[1] : "scripts/r06/bench_cpp_single_call_input.txt", "r");
    if (!f) { std::printf("input file missing\n"); return 2; }
    char line[256];
    if (!std::fgets(line, sizeof line, f)) return 2;   // the comment line
    for (int k = 0; k < model.nq; ++k) if (std::fscanf(f, "%lf", &q[k]) != 1) return 2;
    for (int k = 0; k < 6; ++k) if (std::fscanf(f, "%lf", &b[k]) != 1) return 2;
    std::fclose(f);
  }
  Mat6x6 H = Identity6();
  Motion v{};
  std::vector<Index> ids{static_cast<Index>(model.getJointId("arm_left_7_joint"))};
  std::vector<Mat6x6> Ais{Identity6()};
  std::vector<Vec6> bis{b};
  auto best_of = [&](auto&& f) {
    double best = 1e9;
    for (int r = 0; r < 50; ++r) {
      const auto t0 = clk::now();
      f();
      best = std::min(best, std::chrono::duration<double, std::milli>(clk::now() - t0).count());
    }
    return best;
  };
  solver.Solve(q, H, v, ids, Ais, bis, lb, ub);
  int it = 0;
  bool ok = false;
  const double t_plain = best_of([&] { solver.Solve(); it = solver.get_iter(); ok = solver.get_convergence_status(); });
  const double t_full = best_of([&] { solver.Solve(q, H, v, ids, Ais, bis, lb, ub); it = solver.get_iter(); ok = solver.get_convergence_status(); });
  const double t_tail = best_of([&] { solver.Solve(q, ids[0], Ais[0], bis[0]); it = solver.get_iter(); ok = solver.get_convergence_status(); });
  solver.set_fetch(FirstOrderLoikOptimized::FETCH_NONE);
  const double t_plain_nofetch = best_of([&] { solver.Solve(); });
  std::printf("{\"what\": \"C++ mirror, one Talos-32 problem per call, %d iterations, converged %d; ms per call incl. the data object's members and get_iter() / "
              "get_convergence_status()\", \"Solve()\": %.4f, \"Solve(q, H_ref, v_ref, ids, Ais, bis, lb, ub)\": %.4f, \"Solve(q, c_id, Ai, bi)\": %.4f, "
              "\"Solve() with set_fetch(FETCH_NONE)\": %.4f}\n", it, (int)ok, t_plain, t_full, t_tail, t_plain_nofetch);
  return 0;
}
