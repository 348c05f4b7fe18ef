// Forwarding header: lets sources written against the reference's include path "kernels/decompression/dictionary.cuh" compile unchanged.
#pragma once
#include "giddy/kernels/decompression/dictionary.hpp"
