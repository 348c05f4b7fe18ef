// Forwarding header: lets sources written against the reference's include path "cuda/optional_and_any.hpp" compile unchanged.
#pragma once
#include "giddy/optional_and_any.hpp"
