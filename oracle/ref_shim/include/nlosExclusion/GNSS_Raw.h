// What ROS' message generator would emit for nlosExclusion/msg/GNSS_Raw.msg (the fields, in the order of the .msg file:
// /root/reference/nlosExclusion/msg/GNSS_Raw.msg:1-32) -- generated code is not under /root/reference, so it is restated here
// as a plain struct (oracle/ref_shim, test infrastructure).
#pragma once
#include <cstdint>
#include <string>
namespace nlosExclusion {
struct GNSS_Raw {
    double GNSS_week = 0, GNSS_time = 0, total_sv = 0, prn_satellites_index = 0, pseudorange = 0, raw_pseudorange = 0, carrier_phase = 0, doppler = 0, lamda = 0, snr = 0;
    int64_t LLI = 0, slip = 0;
    double elevation = 0, azimuth = 0, err_tropo = 0, err_iono = 0, sat_clk_err = 0, sat_pos_x = 0, sat_pos_y = 0, sat_pos_z = 0, ttx = 0, vel_x = 0, vel_y = 0, vel_z = 0, dt = 0, ddt = 0, tgd = 0;
    int64_t visable = 0;
    std::string sat_system;
    int64_t visable3DMA = 0;
    double prE3dMA = 0;
};
}
