// sf_solver_f32.hip — the float instantiation of sfi::Solver (host logic + every gfx950 kernel it launches).
#include "sf_solver.hpp"

namespace sfi {
SolverBase* make_solver_f32(const sf_params& p) { return new Solver<float>(p); }
}  // namespace sfi
