// Event pool shared by the multi-stream host code (net.hip: passes of the network executor; maps.hip: kernel maps).
#pragma once
#include "common.h"

namespace osn {
int events_count(const osn_events_t* e);
hipEvent_t events_get(const osn_events_t* e, int i);
}  // namespace osn
