// Forwarding header: lets sources written against the reference's include path "util/Factory.h" compile unchanged.
#pragma once
#include "giddy/util/factory.hpp"
