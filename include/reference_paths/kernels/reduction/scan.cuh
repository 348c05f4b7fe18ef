// forwarding header: the reference's include path for this component (src/kernels/reduction/scan.cuh)
#pragma once
#include "giddy/kernels/reduction/scan.hpp"
