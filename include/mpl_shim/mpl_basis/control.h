/** @file control.h  (mplx shim of <mpl_basis/control.h>): control kinds = union of the Waypoint use_* bits */
#ifndef MPLX_SHIM_CONTROL_H
#define MPLX_SHIM_CONTROL_H
namespace Control {
enum Control {
  NONE = 0,
  VEL = 0b00001,
  ACC = 0b00011,
  JRK = 0b00111,
  SNP = 0b01111,
  VELxYAW = 0b10001,
  ACCxYAW = 0b10011,
  JRKxYAW = 0b10111,
  SNPxYAW = 0b11111
};
}
#endif
