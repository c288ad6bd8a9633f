// ORACLE — TEST INFRASTRUCTURE ONLY.
// Link-time stand-ins for the three networking entry points the reference's core settings code refers to
// (client-side extrapolation worker, src/edyn/networking/extrapolation/extrapolation_worker.cpp — out of scope,
// SURVEY.md §2). They are reachable only when a network client context exists, which the checker never creates.
#include <edyn/networking/extrapolation/extrapolation_worker.hpp>
#include <cstdlib>

namespace edyn {
void extrapolation_worker::set_settings(const edyn::settings &) { std::abort(); }
void extrapolation_worker::set_material_table(const material_mix_table &) { std::abort(); }
void extrapolation_worker::set_registry_operation_context(const registry_operation_context &) { std::abort(); }
}  // namespace edyn
