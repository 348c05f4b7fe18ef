// Forwarding header: lets sources written against the reference's include path "kernel_wrappers/decompression/discard_zero_bytes_fixed.cu" compile unchanged.
#pragma once
#include "giddy/kernel_wrappers/decompression/discard_zero_bytes_fixed.hpp"
