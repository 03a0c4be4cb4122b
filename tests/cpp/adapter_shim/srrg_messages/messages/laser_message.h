// test shim (tests/cpp/adapter_shim/srrg_shim.h): stands in for <srrg_messages/messages/laser_message.h> when compile-checking adapters/srrg/
#pragma once
#include "srrg_shim.h"
