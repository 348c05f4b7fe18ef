// Forwarding header: lets sources written against the reference's include path "kernels/data_layout/gather.cuh" compile unchanged.
#pragma once
#include "giddy/kernels/data_layout/gather.hpp"
