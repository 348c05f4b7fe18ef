// Forwarding header: lets sources written against the reference's include path "kernel_wrappers/decompression/incidence_bitmaps.cu" compile unchanged.
#pragma once
#include "giddy/kernel_wrappers/decompression/incidence_bitmaps.hpp"
