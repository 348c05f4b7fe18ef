// Forwarding header: lets sources written against the reference's include path "util/integer.h" compile unchanged.
#pragma once
#include "giddy/util/integer.hpp"
