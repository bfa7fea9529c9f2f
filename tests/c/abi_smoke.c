/* A C caller of libbodge_hip.so, compiled by tests/test_abi.py with plain gcc against
 * include/bodge_hip.h: the boundary is usable without Python and without C++.
 * Only calls that need no GPU are made (version, device count, argument validation). */
#include <stdio.h>
#include <string.h>

#include "bodge_hip.h"

int main(void) {
    int devices = -1;
    bdg_system* handle = (bdg_system*)0;
    int32_t indptr[2] = {0, 2}; /* claims two blocks ... */
    int32_t indices[1] = {0};
    double data[32] = {0};

    printf("version: %s\n", bdg_version());
    if (bdg_device_count(&devices) != BDG_OK || devices < 0) return 1;
    printf("devices: %d\n", devices);

    /* ... but nnzb = 1: must be refused before anything touches a device */
    int rc = bdg_create(0, 1, 1, indptr, indices, data, &handle);
    printf("bdg_create on a malformed matrix: rc = %d, message = \"%s\"\n", rc, bdg_last_error());
    if (rc == BDG_OK || handle != (bdg_system*)0) return 2;
    if (strstr(bdg_last_error(), "indptr") == (char*)0) return 3;
    if (bdg_destroy((bdg_system*)0) != BDG_OK) return 4; /* destroying nothing is fine */
    return 0;
}
