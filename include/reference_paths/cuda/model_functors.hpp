// Forwarding header: lets sources written against the reference's include path "cuda/model_functors.hpp" compile unchanged.
#pragma once
#include "giddy/functors/model_functors.hpp"
