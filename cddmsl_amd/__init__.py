"""cddmsl_amd: MI355X-native hot path of CDDMSL."""
