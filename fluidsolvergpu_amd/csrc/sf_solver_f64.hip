// sf_solver_f64.hip — the double instantiation of sfi::Solver (host logic + every gfx950 kernel it launches).
#include "sf_solver.hpp"

namespace sfi {
SolverBase* make_solver_f64(const sf_params& p) { return new Solver<double>(p); }
}  // namespace sfi
