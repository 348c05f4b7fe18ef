// Forwarding header: lets sources written against the reference's include path "kernels/decompression/delta.cuh" compile unchanged.
#pragma once
#include "giddy/kernels/decompression/delta.hpp"
