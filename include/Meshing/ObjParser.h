// Forwarding header: lets code written against the reference include layout compile unchanged.
#pragma once
#include "../hpsdf_meshing.hpp"
