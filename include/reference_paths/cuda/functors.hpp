// forwarding header: the reference's include path for this component (src/cuda/functors.hpp)
#pragma once
#include "giddy/functors/functors.hpp"
