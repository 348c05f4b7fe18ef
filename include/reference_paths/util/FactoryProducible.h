// Forwarding header: lets sources written against the reference's include path "util/FactoryProducible.h" compile unchanged.
#pragma once
#include "giddy/util/factory.hpp"
