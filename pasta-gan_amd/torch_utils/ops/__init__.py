# Op front-ends over libpasta_hip.so (C ABI: ../../../include/pasta_hip.h).
