// ORACLE — TEST INFRASTRUCTURE ONLY (see dual.hpp header).
// Universal PC-SAFT model constants, values as published by Gross & Sadowski (2001) and
// Gross & Vrabec (2006) and as held by the reference at feos_torch/pcsaft_pure.py:10-86.
#pragma once

namespace oracle {

static const double A0[7] = {0.91056314451539, 0.63612814494991, 2.68613478913903, -26.5473624914884,
                             97.7592087835073, -159.591540865600, 91.2977740839123};
static const double A1[7] = {-0.30840169182720, 0.18605311591713, -2.50300472586548, 21.4197936296668,
                             -65.2558853303492, 83.3186804808856, -33.7469229297323};
static const double A2[7] = {-0.09061483509767, 0.45278428063920, 0.59627007280101, -1.72418291311787,
                             -4.13021125311661, 13.7766318697211, -8.67284703679646};
static const double B0[7] = {0.72409469413165, 2.23827918609380, -4.00258494846342, -21.00357681484648,
                             26.8556413626615, 206.5513384066188, -355.60235612207947};
static const double B1[7] = {-0.57554980753450, 0.69950955214436, 3.89256733895307, -17.21547164777212,
                             192.6722644652495, -161.8264616487648, -165.2076934555607};
static const double B2[7] = {0.09768831158356, -0.25575749816100, -9.15585615297321, 20.64207597439724,
                             -38.80443005206285, 93.6267740770146, -29.66690558514725};
static const double AD[5][3] = {{0.30435038064, 0.95346405973, -1.16100802773},
                                {-0.13585877707, -1.83963831920, 4.52586067320},
                                {1.44933285154, 2.01311801180, 0.97512223853},
                                {0.35569769252, -7.37249576667, -12.2810377713},
                                {-2.06533084541, 8.23741345333, 5.93975747420}};
static const double BD[5][3] = {{0.21879385627, -0.58731641193, 3.48695755800},
                                {-1.18964307357, 1.24891317047, -14.9159739347},
                                {1.16268885692, -0.50852797392, 15.3720218600},
                                {0.0, 0.0, 0.0},
                                {0.0, 0.0, 0.0}};
static const double CD[4][3] = {{-0.06467735252, -0.95208758351, -0.62609792333},
                                {0.19758818347, 2.99242575222, 1.29246858189},
                                {-0.80875619458, -2.38026356489, 1.65427830900},
                                {0.69028490492, -0.27012609786, -3.43967436378}};

static const double PI = 3.14159265358979323846;  // numpy.pi
// si_units values used by the reference (SI base units as plain floats)
static const double KB = 1.380649e-23;    // J/K
static const double NAV = 6.02214076e23;  // 1/mol
static const double ANGSTROM = 1e-10;     // m
// feos_torch/pcsaft_pure.py:97-98  1e-19 * (JOULE / KELVIN / KB)
static const double MU2_UNIT = 1e-19 / KB;
// feos_torch/pcsaft_pure.py:215   KB*KELVIN/ANGSTROM**3/PASCAL  (reduced pressure * T -> Pa)
static const double P_UNIT = KB / (ANGSTROM * ANGSTROM * ANGSTROM);
// feos_torch/pcsaft_pure.py:199   (KILO*MOL/METER**3) * (NAV*ANGSTROM**3)  (A^-3 -> kmol/m3)
static const double RHO_UNIT = 1e3 * (NAV * (ANGSTROM * ANGSTROM * ANGSTROM));

}  // namespace oracle
