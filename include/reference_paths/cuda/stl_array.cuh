// Forwarding header: lets sources written against the reference's include path "cuda/stl_array.cuh" compile unchanged.
#pragma once
#include "giddy/array.hpp"
