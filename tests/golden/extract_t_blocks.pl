#!/usr/bin/env perl
# Extract the data blocks of the reference's Test::Base files (t/*.t) into JSONL.
#
# Test::Base itself is not installed here, so this re-implements just the block
# syntax the reference's suite uses (t/SRegex.pm:27-75 reads: re, s, flags, cap,
# match_id, temp_cap, err, err_like, fatal, no_match, SKIP):
#
#   === <name>                       block delimiter
#   --- <section> [filters][: value] single-line value (trimmed), or
#   --- <section> [filters]          multi-line value up to the next delimiter
#
# default filters are norm+trim, explicit ones used by the suite are `eval`
# and `chop`.  Byte strings are emitted hex-encoded so that nothing is lost in
# JSON.  Only DATA is extracted (inputs + the few explicit expectations); the
# expected outputs proper come from running the reference (make_goldens.py).
#
# usage: perl extract_t_blocks.pl /root/reference/t > blocks.jsonl
use strict;
use warnings;
use JSON::PP;

my $dir = shift // die "usage: $0 <reference t/ dir>\n";
my $json = JSON::PP->new->canonical(1);

sub to_bytes {
    my ($s) = @_;
    return undef unless defined $s;
    # what perl's print/exec would hand to the OS: wide strings go out as UTF-8
    if ($s =~ /[^\x00-\xff]/) {
        utf8::encode($s);
    }
    return $s;
}

sub hexs { my $b = to_bytes($_[0]); return unpack("H*", $b); }

sub f_norm { my $t = shift; $t =~ s/\015\012/\n/g; $t =~ s/\r/\n/g; return $t; }
sub f_trim {
    my $t = shift;
    $t =~ s/\A([ \t]*\n)+//;
    $t =~ s/(?<=\n)\s*\z//g;
    return $t;
}

for my $file (sort glob("$dir/*.t")) {
    open my $in, "<", $file or die "$file: $!";
    binmode $in;
    local $/;
    my $src = <$in>;
    close $in;
    $src =~ s/\A.*?^__DATA__\n//ms or next;
    (my $base = $file) =~ s{.*/}{};

    my @hunks = split /^(?====[ \t])/m, $src;
    for my $hunk (@hunks) {
        next unless $hunk =~ s/\A===[ \t]*(.*)\s+//;
        my $name = $1;
        my @parts = split /^--- +\(?(\w+)\)? *(.*)?\n/m, $hunk;
        shift @parts;    # description
        my %blk = (file => $base, name => $name);
        while (my ($type, $filters, $value) = splice(@parts, 0, 3)) {
            $value = '' unless defined $value;
            $filters = '' unless defined $filters;
            my $oneline = 0;
            if ($filters =~ /:(\s|\z)/) {
                ($filters, $value) = split /\s*:(?:\s+|\z)/, $filters, 2;
                $value = '' unless defined $value;
                $value =~ s/^\s*(.*?)\s*$/$1/;
                $oneline = 1;
            }
            $filters = '' unless defined $filters;
            $value = f_trim(f_norm($value));
            for my $f (split ' ', $filters) {
                if ($f eq 'eval') {
                    my $v = eval $value;
                    die "$base $name: eval failed for $type: $@" if $@;
                    $value = $v;
                } elsif ($f eq 'chop') {
                    chop $value;
                } else {
                    die "$base $name: unknown filter $f";
                }
            }
            if ($type eq 're') {
                my @res = ref $value eq 'ARRAY' ? @$value : ($value);
                $blk{re} = [ map { hexs($_) } @res ];
                $blk{multi} = ref $value eq 'ARRAY' ? JSON::PP::true : JSON::PP::false;
            } elsif ($type eq 's') {
                $blk{s} = hexs($value);
            } elsif ($type eq 'flags') {
                $blk{flags} = "$value";
            } elsif ($type eq 'cap' && ref $value) {
                $blk{cap_like} = "$value";
            } elsif ($type =~ /^(cap|match_id|temp_cap|err|err_like)$/) {
                $blk{$type} = "$value";
            } elsif ($type =~ /^(fatal|no_match|SKIP)$/) {
                $blk{lc $type} = JSON::PP::true;
            } else {
                die "$base $name: unknown section $type";
            }
        }
        die "$base $name: no re/s" unless defined $blk{re} && defined $blk{s};
        print $json->encode(\%blk), "\n";
    }
}
