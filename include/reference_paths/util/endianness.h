// Forwarding header: lets sources written against the reference's include path "util/endianness.h" compile unchanged.
#pragma once
#include "giddy/util/endianness.hpp"
