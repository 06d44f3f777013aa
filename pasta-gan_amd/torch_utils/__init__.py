# MI355X-native replacement for the PASTA-GAN hot path; see ../../DESIGN.md.
