// Forwarding header: lets sources written against the reference's include path "kernels/decompression/incidence_bitmaps.cuh" compile unchanged.
#pragma once
#include "giddy/kernels/decompression/incidence_bitmaps.hpp"
