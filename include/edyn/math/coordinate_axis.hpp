// Forwarding header of the MI355X stepper shim: the reference splits its API over many headers, the shim keeps it in one.
// Code that includes <edyn/math/coordinate_axis.hpp> (as code written against the reference does) gets the shim's declarations.
#ifndef EDYN_HIP_FWD_MATH_COORDINATE_AXIS_HPP
#define EDYN_HIP_FWD_MATH_COORDINATE_AXIS_HPP
#include <edyn/edyn.hpp>
#endif
