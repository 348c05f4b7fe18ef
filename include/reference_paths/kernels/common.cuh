// Forwarding header: lets sources written against the reference's include path "kernels/common.cuh" compile unchanged.
#pragma once
#include "giddy/common.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
