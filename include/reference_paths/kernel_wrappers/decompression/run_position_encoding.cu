// Forwarding header: lets sources written against the reference's include path "kernel_wrappers/decompression/run_position_encoding.cu" compile unchanged.
#pragma once
#include "giddy/kernel_wrappers/decompression/run_position_encoding.hpp"
