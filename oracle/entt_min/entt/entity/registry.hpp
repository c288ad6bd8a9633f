// ORACLE — TEST INFRASTRUCTURE ONLY. Forwarding header: see entt/entt_min.hpp.
#include "../entt_min.hpp"
