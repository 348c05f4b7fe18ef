// Forwarding header: lets sources written against the reference's include path "kernels/resolve_launch_configuration.h" compile unchanged.
#pragma once
#include "giddy/kernels/resolve_launch_configuration.hpp"
