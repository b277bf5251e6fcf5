"""MI355X-native implementation of phylonium's anchor + pairwise-compare hot path."""
