// forwarding header: the reference's io.hpp is split here into config.hpp (SimConfig, YAML/CLI)
// and snapshot.hpp (NetCDF output)
#pragma once
#include "config.hpp"
#include "snapshot.hpp"
