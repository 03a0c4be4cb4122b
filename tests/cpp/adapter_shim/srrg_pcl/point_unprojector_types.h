// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg_pcl/point_unprojector_types.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
