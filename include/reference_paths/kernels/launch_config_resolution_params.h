// Forwarding header: lets sources written against the reference's include path "kernels/launch_config_resolution_params.h" compile unchanged.
#pragma once
#include "giddy/kernels/launch_config_resolution_params.hpp"
