// Forwarding header of the MI355X stepper shim: the reference splits its API over many headers, the shim keeps it in one.
// Code that includes <edyn/comp/present_position.hpp> (as code written against the reference does) gets the shim's declarations.
#ifndef EDYN_HIP_FWD_COMP_PRESENT_POSITION_HPP
#define EDYN_HIP_FWD_COMP_PRESENT_POSITION_HPP
#include <edyn/edyn.hpp>
#endif
