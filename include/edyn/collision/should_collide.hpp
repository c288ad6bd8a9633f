// Forwarding header of the MI355X stepper shim: the reference splits its API over many headers, the shim keeps it in one.
// Code that includes <edyn/collision/should_collide.hpp> (as code written against the reference does) gets the shim's declarations.
#ifndef EDYN_HIP_FWD_COLLISION_SHOULD_COLLIDE_HPP
#define EDYN_HIP_FWD_COLLISION_SHOULD_COLLIDE_HPP
#include <edyn/edyn.hpp>
#endif
