// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg_pcl/normal_computator.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
