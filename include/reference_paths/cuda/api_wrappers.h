// Forwarding header: lets sources written against the reference's include path "cuda/api_wrappers.h" compile unchanged.
#pragma once
#include "giddy/api/device.hpp"
#include "giddy/api/types.hpp"
